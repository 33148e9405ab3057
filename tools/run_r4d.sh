TAG=${TAG:-r4d}
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "winograd_wgrad or conv_full_size_properties or conv_oracle" -p no:cacheprovider > gpurun_out/pytest_${TAG}.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_${TAG}.log | tail -3
python tools/conv_bench.py --only wgrad --ab CPG_WW_SHARE=0,- --iters 10 --reps 5 > gpurun_out/ab_${TAG}_share.txt 2>&1; cat gpurun_out/ab_${TAG}_share.txt
