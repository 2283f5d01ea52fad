# Idle time between consecutive decode-attention launches, from a kernel trace of a short run of the default workload.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/gaps
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof/trace -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-profile --steps 1 --warmup 0 --max-tokens 24 > $O/bench.json 2> $O/trace.err; rc=$?
echo "rc=$rc"
[ $rc -eq 0 ] || { tail -3 $O/trace.err; exit 1; }
python3 $R/scripts/trace_gaps.py /tmp/prof/trace decode_attn | tee $O/gaps.txt
