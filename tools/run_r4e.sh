TAG=${TAG:-r4e}
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "byte_mask or resnet or fused_bn or bn_add" -p no:cacheprovider > gpurun_out/pytest_${TAG}.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_${TAG}.log | tail -3
for i in 1 2; do
python bench.py --arch resnet50 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260
python tools/net_bench.py --arch resnet50 --steps 10 2>&1 | tail -1
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --optin-steps 0 2>&1 | tail -1 | cut -c1-260
