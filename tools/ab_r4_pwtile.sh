# pointwise 128-row tile: 4 x 1 waves / 224 pixels (CPG_PW_TILE=0) against 2 x 2 waves / 256 pixels (1): per shape and whole ResNet-50 step
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "pointwise or conv_golden or conv_oracle or epilogue_bn_statistics or resnet" -p no:cacheprovider 2>&1 | tail -1
CPG_PW_TILE=1 timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "pointwise or conv_golden or conv_oracle or epilogue_bn_statistics or resnet or skip_gradient or linear" -p no:cacheprovider 2>&1 | tail -1
for v in 0 1 0 1; do echo "CPG_PW_TILE=$v"; CPG_PW_TILE=$v python tools/generic_bench.py --iters 10 --only 1x1 2>&1 | grep -v amdgpu.ids; done
for v in 0 1 0 1; do echo -n "CPG_PW_TILE=$v  "; CPG_PW_TILE=$v python tools/net_bench.py --arch resnet50 --steps 10 2>&1 | tail -1; done
