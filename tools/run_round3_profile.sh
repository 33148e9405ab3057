# one GPU-box call (round 3): tests, smoke, the bench line of the three topologies (headline = vgg16), rocprofv3 kernel summaries of the
# driver's command per topology.  TAG names the outputs under gpurun_out/.
TAG=${TAG:-r3a}
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_${TAG}.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_${TAG}.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/bench_${TAG}.log 2>&1; tail -1 gpurun_out/bench_${TAG}.log | cut -c1-200
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_k20.log 2>&1; tail -1 gpurun_out/bench_${TAG}_k20.log | cut -c1-200
for a in resnet50 spherenet20; do
  python bench.py --arch $a --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_$a.log 2>&1; tail -1 gpurun_out/bench_${TAG}_$a.log | cut -c1-200
  python tools/net_bench.py --arch $a --steps 10 2>&1 | tail -1 | tee -a gpurun_out/net_${TAG}.txt
done
python tools/generic_bench.py --iters 5 > gpurun_out/generic_${TAG}.txt 2>&1
python tools/conv_bench.py --iters 5 > gpurun_out/conv_bench_${TAG}.txt 2>&1; tail -3 gpurun_out/conv_bench_${TAG}.txt
cd /tmp && export TMPDIR=/tmp
for a in vgg16 resnet50 spherenet20; do
  rm -rf $R/gpurun_out/prof_${TAG}_$a
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_$a -o run -- python $R/bench.py --arch $a --steps 20 --warmup 5 --no-cpu-baseline --optin-steps 0 > $R/gpurun_out/prof_${TAG}_$a.log 2>&1
  db=$(find $R/gpurun_out/prof_${TAG}_$a -name '*.db' | head -1)
  python $R/tools/rocprof_summary.py $db 50 > $R/gpurun_out/summary_${TAG}_$a.md 2>&1
  rm -rf $R/gpurun_out/prof_${TAG}_$a
done
# HBM traffic of the bench's own launch mix: FETCH_SIZE and WRITE_SIZE in separate counter passes (kernel trace only)
for a in vgg16 resnet50 spherenet20; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/btraffic/${a}_$c -o run --output-format csv -- python $R/bench.py --arch $a --steps 20 --warmup 1 --no-cpu-baseline --optin-steps 0 > $R/gpurun_out/btraffic_${a}_$c.log 2>&1
  done
done
python $R/tools/bench_traffic.py $R/gpurun_out/btraffic > $R/gpurun_out/traffic_${TAG}.json 2> $R/gpurun_out/traffic_${TAG}.err
rm -rf $R/gpurun_out/btraffic
