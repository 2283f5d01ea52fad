# HBM counters of the log-mel kernel (one counter per pass) on the stand-alone stage, 1024 clips.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_t
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --kernel-include-regex mel_kernel -d /tmp/prof/mel_$c -o p --output-format csv -- python3 $R/bench.py --workload mel --batch 1024 --steps 1 --warmup 0 > $O/mel_$c.json 2> $O/mel_$c.err; rc=$?
  echo "pmc $c rc=$rc"
  [ $rc -eq 0 ] || { tail -3 $O/mel_$c.err; exit 1; }
done
F=$(find /tmp/prof/mel_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find /tmp/prof/mel_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 $R/scripts/pmc_summary.py $F $W mel_kernel "bench.py --workload mel --batch 1024: 1024 x 30 s clips, device buffers, rows stored scaled with per-cell minima (round-3 kernel)" > $O/pmc_mel_b1024.json || exit 1
cat $O/pmc_mel_b1024.json
