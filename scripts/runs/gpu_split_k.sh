# Projection plans on the 96 CUs beside the attention for the other model / weight-format combinations,
# then the 0.6B model at 2048 clips: one stream against the split decode.
set -o pipefail
O=gpurun_out/split_k
mkdir -p $O
for pw in "1.7b q4" "0.6b bf16" "0.6b q4"; do
  set -- $pw
  timeout -k 10 400 python scripts/plan_table.py --preset $1 --weights $2 --rows 1024,2048,4096 --att-cus 160 > $O/plan_$1_$2_cus96.json 2> $O/plan_$1_$2.err || { echo "plan $pw failed"; tail -5 $O/plan_$1_$2.err; exit 1; }
  tail -1 $O/plan_$1_$2.err
done
