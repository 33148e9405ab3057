# one GPU-box call: tests, smoke, bench (default = full 220-step cycle, and the driver's 20-step form), conv bench,
# rocprofv3 kernel summary of the driver's command, PMC traffic passes.  TAG names the outputs under gpurun_out/.
TAG=${TAG:-r2a}
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_${TAG}.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_${TAG}.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/bench_${TAG}.log 2>&1; tail -1 gpurun_out/bench_${TAG}.log | cut -c1-300
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_k20.log 2>&1; tail -1 gpurun_out/bench_${TAG}_k20.log | cut -c1-300
python tools/conv_bench.py --iters 5 > gpurun_out/conv_bench_${TAG}.txt 2>&1; tail -3 gpurun_out/conv_bench_${TAG}.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${TAG} $R/gpurun_out/traffic_${TAG}
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG} -o run -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --optin-steps 0 > $R/gpurun_out/prof_${TAG}.log 2>&1
for f in fwd dgrad wgrad; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/traffic_${TAG}/${f}_$c -o run --output-format csv -- python $R/tools/conv_bench.py --pmc-pass --only $f > /dev/null 2>&1
done; done
ls $R/gpurun_out/traffic_${TAG}
