# round 5: (1) A/B of the round-4 Winograd kernels vs the any-channel-count ones on the width-1.0 layers (same box, interleaved);
# (2) the grown network's layers (raw multiplier 1.5), default plan and one-wave kernel; (3) the grown bench line
for i in 1 2; do for L in cpg_amd/lib/exp/libcpg_hip_r4.so cpg_amd/lib/libcpg_hip.so; do echo "== $L"; CPG_HIP_LIB=$PWD/$L python tools/conv_bench.py --only fwdstats,dgrad,wgrad --iters 10 2>&1 | grep -E "TOTAL|f3 |f27|f34"; done; done
echo "== grown layers"
python tools/conv_bench.py --width-multiplier 1.5 --only fwdstats,dgrad,wgrad --iters 10
echo "== grown layers, k_wg1 forced"
python tools/conv_bench.py --width-multiplier 1.5 --only fwdstats,dgrad --iters 10 --ab CPG_WINO_KERNEL=-,wave,64
echo "== grown layers, direct kernels"
CPG_NO_WINO=1 python tools/conv_bench.py --width-multiplier 1.5 --only fwdstats,dgrad,wgrad --iters 5
python bench.py --width-multiplier 1.5 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5b_bench_grown.log 2>&1; tail -1 gpurun_out/r5b_bench_grown.log | cut -c1-400
