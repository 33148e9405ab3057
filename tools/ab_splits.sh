# A/B of the weight-gradient split counts (partial-sum volume vs units per slot) on the whole train step
for a in spherenet20 resnet50 vgg16; do
  echo "== $a"
  for v in "X=0" "CPG_WW_UNITS=2" "CPG_WW_UNITS=4" "CPG_PWW_BPC=1" "CPG_PWW_BPC=3" "CPG_C3W_BPC=1" "CPG_C3W_BPC=3" "X=1"; do
    echo -n "$v  "; env $v python tools/net_bench.py --arch $a --steps 10 2>&1 | tail -1
  done
done
