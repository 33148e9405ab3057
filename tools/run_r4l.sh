for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --optin-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vgg16 t1', d['value'], d['ms_per_step'], d['phases'])"
done
python bench.py --task 2 --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vgg16 t2', d['value'], d['ms_per_step'], d['task2']['task1_ms_per_step'], d['task2']['task2_over_task1'], d['phases'])"
python bench.py --arch resnet50 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('r50 t1', d['value'], d['ms_per_step'], d['phases'])"
python bench.py --arch resnet50 --task 2 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('r50 t2', d['value'], d['ms_per_step'], d['task2']['task1_ms_per_step'], d['task2']['task2_over_task1'], d['phases'])"
python bench.py --arch spherenet20 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sph t1', d['value'], d['ms_per_step'], d['phases'])"
