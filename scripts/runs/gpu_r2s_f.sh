# stdout of bench.py holds exactly one line (the JSON) also when RCCL announces itself; one-rank rehearsal of the exchange.
set -o pipefail
O=gpurun_out/r2s_f
mkdir -p $O
timeout -k 10 300 python bench.py --rehearse-collectives --batch 8 --max-tokens 8 --no-cpu-baseline > $O/rehearse_nccl.json 2> $O/rehearse_nccl.err || { tail -20 $O/rehearse_nccl.err; exit 1; }
wc -l $O/rehearse_nccl.json; cut -c1-120 $O/rehearse_nccl.json; grep -c "RCCL version" $O/rehearse_nccl.err
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --backend gloo --batch 8 --max-tokens 8 --no-cpu-baseline > $O/launcher2.json 2> $O/launcher2.err || { tail -20 $O/launcher2.err; exit 1; }
wc -l $O/launcher2.json; cut -c1-160 $O/launcher2.json
