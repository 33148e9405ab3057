TAG=${TAG:-r4h}
timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "winograd or conv_full_size_properties or conv_oracle or epilogue_bn_statistics" -p no:cacheprovider > gpurun_out/pytest_${TAG}.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_${TAG}.log | tail -3
python tools/conv_bench.py --only fwdstats,dgrad --ab CPG_WG3_SHARE=0,- --iters 10 --reps 5 > gpurun_out/ab_${TAG}_wg3share.txt 2>&1; cat gpurun_out/ab_${TAG}_wg3share.txt
