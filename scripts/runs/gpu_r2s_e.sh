# The multi-GPU exchange (RCCL all_gather / all_reduce / barrier) rehearsed through a process group of one rank.
set -o pipefail
O=gpurun_out/r2s_e
mkdir -p $O
timeout -k 10 300 python bench.py --rehearse-collectives --batch 8 --max-tokens 8 --no-cpu-baseline > $O/rehearse_nccl.json 2> $O/rehearse_nccl.err || { tail -20 $O/rehearse_nccl.err; exit 1; }
cut -c1-400 $O/rehearse_nccl.json
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --rehearse-collectives --batch 8 --max-tokens 8 --no-cpu-baseline > $O/rehearse_torchrun.json 2> $O/rehearse_torchrun.err || { tail -20 $O/rehearse_torchrun.err; exit 1; }
cut -c1-200 $O/rehearse_torchrun.json
