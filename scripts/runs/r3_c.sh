# Round 3, run c: (1) the decode loop's event chain between CU-masked streams with destroy / re-create between rounds,
# then with the streams kept; (2) the whole GPU suite; (3) the launcher's 2-rank rehearsal line; (4) a host that ends
# without q3asr_destroy under rocprofv3, with and without the library's exit handler; (5) the default bench line.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3_c
mkdir -p $O
cd $R



timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a $O/tests.log; tail -5 $O/tests.log
cp gpurun_out/parity_report.txt $O/ 2>/dev/null
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --backend gloo --batch 8 --max-tokens 8 --no-cpu-baseline > $O/launcher_2ranks.json 2> $O/launcher_2ranks.err || { tail -5 $O/launcher_2ranks.err; exit 1; }
cut -c1-400 $O/launcher_2ranks.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof/leak_hook -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 --batch 256 --max-tokens 16 --leak-model > $O/leak_with_hook.json 2> $O/leak_with_hook.err
echo "leak with exit handler rc=$?" | tee $O/leak_rc.txt
export Q3ASR_NO_EXIT_HOOK=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof/leak_nohook -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 --batch 256 --max-tokens 16 --leak-model > $O/leak_no_hook.json 2> $O/leak_no_hook.err
echo "leak without exit handler rc=$?" | tee -a $O/leak_rc.txt
unset Q3ASR_NO_EXIT_HOOK
cd $R
timeout -k 10 600 python bench.py --steps 3 --warmup 1 > $O/bench_b2048.json 2> $O/bench_b2048.err || { tail -5 $O/bench_b2048.err; exit 1; }
cut -c1-300 $O/bench_b2048.json
