python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "linear or two_task or trajectory" 2>&1 | tail -3
for i in 1 2; do python bench.py --task 2 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['task2']['task1_ms_per_step'], d['task2']['task2_over_task1'], {k:(v['avg_launch_ms'],v['hbm_floor_ms_per_launch'],v['frac_of_dense_peak_executed']) for k,v in d['kernel_families'].items() if 'linear' in k})"; done
COMMIT=1822428 LIMIT=1500 bash tools/run_traffic.sh
