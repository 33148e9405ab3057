# A/B of two library builds through CPG_HIP_LIB on one box: conv_bench fwdstats + dgrad per layer
for i in 1 2; do for L in $1 $2; do echo "== $L"; CPG_HIP_LIB=$PWD/$L python tools/conv_bench.py --only fwdstats,dgrad --iters 10 2>&1 | grep -E "TOTAL|f3|f27"; done; done
