R=$PWD
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/bench_r1k.log 2>&1; tail -1 gpurun_out/bench_r1k.log | cut -c1-400
python tools/conv_bench.py --iters 5 > gpurun_out/conv_bench_k.txt 2>&1; tail -3 gpurun_out/conv_bench_k.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r1k $R/gpurun_out/traffic
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1k -o run -- python $R/bench.py > $R/gpurun_out/prof_r1k.log 2>&1
for f in fwd dgrad wgrad; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/traffic/${f}_$c -o run --output-format csv -- python $R/tools/conv_bench.py --pmc-pass --only $f > /dev/null 2>&1
done; done
ls $R/gpurun_out/traffic
