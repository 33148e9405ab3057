# round 4: tests + the task-2 line again (after the prune-mode Adam shortcut and the double-precision Adam hyper-parameters)
TAG=${TAG:-r4b}
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_${TAG}.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_${TAG}.log | tail -3
export CPG_BENCH_DETAIL=1
python bench.py --task 2 --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_task2.log 2>&1; tail -1 gpurun_out/bench_${TAG}_task2.log | cut -c1-300
python bench.py --task 2 --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_task2_again.log 2>&1; tail -1 gpurun_out/bench_${TAG}_task2_again.log | cut -c1-300
