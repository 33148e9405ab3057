TAG=${TAG:-r4j}
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_${TAG}.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_${TAG}.log | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --optin-steps 0 2>&1 | tail -1 | cut -c1-300
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --optin-steps 0 2>&1 | tail -1 | cut -c1-300
for a in resnet50 spherenet20; do python bench.py --arch $a --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230; done
python bench.py --task 2 --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['task2'])"
