# one GPU-box call: train-step time + rocprofv3 kernel summary of the ResNet-50 / SphereNet-20 topologies (configs 4 / 5)
TAG=${TAG:-r3a}
R=$PWD
python tools/net_bench.py --arch resnet50 --steps 10 2>&1 | tail -1 | tee gpurun_out/net_${TAG}.txt
python tools/net_bench.py --arch spherenet20 --steps 10 2>&1 | tail -1 | tee -a gpurun_out/net_${TAG}.txt
python tools/generic_bench.py --iters 5 2>&1 | tee gpurun_out/generic_${TAG}.txt | tail -20
cd /tmp && export TMPDIR=/tmp
for a in resnet50 spherenet20; do
  rm -rf $R/gpurun_out/prof_${TAG}_$a
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_$a -o run -- python $R/tools/net_bench.py --arch $a --steps 5 > $R/gpurun_out/prof_${TAG}_$a.log 2>&1
  db=$(find $R/gpurun_out/prof_${TAG}_$a -name '*.db' | head -1)
  python $R/tools/rocprof_summary.py $db 45 > $R/gpurun_out/summary_${TAG}_$a.md 2>&1
  rm -rf $R/gpurun_out/prof_${TAG}_$a
done
