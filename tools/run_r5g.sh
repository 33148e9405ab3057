# round 5: the Winograd tail pieces: parity, then SphereNet-20 / VGG16 / ResNet-50 with and without them (interleaved on one box)
python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "tail_pieces or winograd or conv_full_size or conv_oracle or inference_epilogue" 2>&1 | tail -8
for i in 1 2; do for t in 0 1; do
  echo "== CPG_WINO_TAIL=$t"
  CPG_WINO_TAIL=$t python bench.py --arch spherenet20 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sph', d['value'], d['ms_per_step'], {k:(v['ms'],v['frac_of_dense_peak_executed']) for k,v in d['kernel_families'].items()})"
done; done
for t in 0 1; do
  echo "== CPG_WINO_TAIL=$t"
  CPG_WINO_TAIL=$t python bench.py --steps 20 --warmup 5 --no-cpu-baseline --optin-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vgg', d['value'], d['ms_per_step'], {k:(v['ms'],v['frac_of_dense_peak_executed']) for k,v in d['kernel_families'].items()})"
  CPG_WINO_TAIL=$t python bench.py --arch resnet50 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('r50', d['value'], d['ms_per_step'], {k:(v['ms'],v['frac_of_dense_peak_executed']) for k,v in d['kernel_families'].items()})"
done
python bench.py --arch spherenet20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sph 220', d['value'], d['ms_per_step'])"
