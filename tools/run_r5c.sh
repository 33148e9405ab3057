# round 5: the any-channel-count kernels after the scalar-offset fixes: parity subset, A/B vs round 4 on the width-1.0 layers, the grown layers, the grown bench
python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "conv_oracle or conv_full_size or winograd or inference_epilogue or full_width" 2>&1 | tail -15
for i in 1 2; do for L in cpg_amd/lib/exp/libcpg_hip_r4.so cpg_amd/lib/libcpg_hip.so; do echo "== $L"; CPG_HIP_LIB=$PWD/$L python tools/conv_bench.py --only fwdstats,dgrad,wgrad --iters 10 2>&1 | grep -E "TOTAL|f3 |f27|f34"; done; done
echo "== grown layers"
python tools/conv_bench.py --width-multiplier 1.5 --only fwdstats,dgrad,wgrad --iters 10
python bench.py --width-multiplier 1.5 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5c_bench_grown.log 2>&1; tail -1 gpurun_out/r5c_bench_grown.log | cut -c1-300
