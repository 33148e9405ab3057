# round 4, first GPU call: tests + smoke, the task-2 cycle (bench.py --task 2) with its rocprofv3 kernel summary, and the per-GPU compute of
# the reference's unscaled-batch data-parallel split (bench.py --batch 128 / 64 / 32).  TAG names the outputs under gpurun_out/.
TAG=${TAG:-r4a}
R=$PWD
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_${TAG}.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_${TAG}.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
export CPG_BENCH_DETAIL=1
python bench.py --task 2 --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_task2.log 2>&1; tail -1 gpurun_out/bench_${TAG}_task2.log | cut -c1-300
for b in 128 64 32; do
  python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --optin-steps 0 > gpurun_out/bench_${TAG}_b$b.log 2>&1; tail -1 gpurun_out/bench_${TAG}_b$b.log | cut -c1-200
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --optin-steps 0 > gpurun_out/bench_${TAG}_b256.log 2>&1; tail -1 gpurun_out/bench_${TAG}_b256.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
for cfg in "task2:--task 2" "b32:--batch 32" "b64:--batch 64"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  rm -rf $R/gpurun_out/prof_${TAG}_$name
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_$name -o run -- python $R/bench.py $flags --steps 20 --warmup 5 --no-cpu-baseline --optin-steps 0 > $R/gpurun_out/prof_${TAG}_$name.log 2>&1
  db=$(find $R/gpurun_out/prof_${TAG}_$name -name '*.db' | head -1)
  python $R/tools/rocprof_summary.py $db 60 > $R/gpurun_out/summary_${TAG}_$name.md 2>&1
  rm -rf $R/gpurun_out/prof_${TAG}_$name
done
