# round 5: pointwise input gradient with the identity add -- addend loads pipelined across fragments: A/B of two library builds on ResNet-50
for i in 1 2 3; do for L in cpg_amd/lib/exp/libcpg_hip_pwbefore.so cpg_amd/lib/libcpg_hip.so; do
  echo "== $L"
  CPG_HIP_LIB=$PWD/$L python bench.py --arch resnet50 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('r50', d['value'], d['ms_per_step'], {k:(v['ms'],v['frac_of_dense_peak_executed']) for k,v in d['kernel_families'].items()})"
done; done
python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "skip_gradient or resnet or pointwise" 2>&1 | tail -3
