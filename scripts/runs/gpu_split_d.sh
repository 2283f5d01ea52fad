# Decode projection candidates measured on the stream confined to the CUs the attention leaves free.
set -o pipefail
O=gpurun_out/split_d
mkdir -p $O
for cu in 128 144 160; do
  timeout -k 10 400 python scripts/plan_table.py --preset 1.7b --weights bf16 --rows 1024,2048,4096 --att-cus $cu > $O/plan_1.7b_bf16_att$cu.json 2> $O/plan_att$cu.err || { echo "plan att_cus=$cu failed"; tail -5 $O/plan_att$cu.err; exit 1; }
  tail -1 $O/plan_att$cu.err
done
