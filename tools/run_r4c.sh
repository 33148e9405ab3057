TAG=${TAG:-r4c}
export CPG_BENCH_DETAIL=1
for a in resnet50 spherenet20; do
  python bench.py --arch $a --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_${TAG}_$a.log 2>&1; tail -1 gpurun_out/bench_${TAG}_$a.log | cut -c1-200
done
python tools/conv_bench.py --only fwdstats,fwd --layers f3,f7 --ab CPG_WINO_KERNEL=-,64 --iters 10 > gpurun_out/ab_${TAG}_wg3_64.txt 2>&1; cat gpurun_out/ab_${TAG}_wg3_64.txt
