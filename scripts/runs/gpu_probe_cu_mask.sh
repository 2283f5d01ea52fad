# CU-mask probe: mapping of mask bits, then read bandwidth against the CU count (alone and beside an MFMA kernel).
O=gpurun_out/probe_cu_mask.txt
: > $O
for a in "0 255" "0 127" "0 95" "0 31" "0 7" "0 255 8" "0 255 2" "8 15"; do
  timeout -k 10 60 ./build_probe/probe_cu_mask where $a >> $O 2>&1 || { echo "rc=$? for where $a" >> $O; }
done
timeout -k 10 200 ./build_probe/probe_cu_mask bw >> $O 2>&1; echo "bw rc=$?" >> $O
cat $O
