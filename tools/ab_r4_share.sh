# headline A/B of round 4's shared-transform kernels on one box: alternate the round-3 configuration (CPG_WG3_SHARE=0 CPG_WW_SHARE=0) and the default
for i in 1 2 3; do
  for v in "CPG_WG3_SHARE=0 CPG_WW_SHARE=0" "X=1"; do
    echo -n "$v  "; env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --optin-steps 0 --no-kernel-clock 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
