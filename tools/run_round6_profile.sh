# one GPU-box call (round 6): tests, smoke, the bench lines (headline = vgg16 task 1; task 2; the reference's unscaled-batch split 128 / 64 / 32;
# the other two topologies; the grown network; the 3-task sequence), rocprofv3 kernel summaries (headline, task 2, batch 32, the other topologies), FETCH_SIZE / WRITE_SIZE
# passes over the bench itself.  TAG names the outputs under gpurun_out/; COMMIT is stamped into the traffic file.
TAG=${TAG:-r6a}
R=$PWD
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_${TAG}.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_${TAG}.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/smoke_${TAG}.txt
python bench.py --no-other-workloads > gpurun_out/bench_${TAG}.log 2>&1; tail -1 gpurun_out/bench_${TAG}.log | cut -c1-200
# the DRIVER's command, exactly (full CPU baseline, the other single-GPU workloads in child processes): its wall time is part of the evidence
SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_k20.log 2>&1; echo "driver command wall: $SECONDS s" | tee gpurun_out/bench_${TAG}_k20.wall; tail -1 gpurun_out/bench_${TAG}_k20.log | cut -c1-200
python bench.py --task 2 --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_task2.log 2>&1; tail -1 gpurun_out/bench_${TAG}_task2.log | cut -c1-200
python bench.py --task 2 > gpurun_out/bench_${TAG}_task2_k220.log 2>&1; tail -1 gpurun_out/bench_${TAG}_task2_k220.log | cut -c1-200
export CPG_BENCH_DETAIL=1
for b in 128 64 32; do
  python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --optin-steps 0 > gpurun_out/bench_${TAG}_b$b.log 2>&1; tail -1 gpurun_out/bench_${TAG}_b$b.log | cut -c1-200
done
unset CPG_BENCH_DETAIL
for a in resnet50 spherenet20; do
  python bench.py --arch $a --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_$a.log 2>&1; tail -1 gpurun_out/bench_${TAG}_$a.log | cut -c1-200
  python tools/net_bench.py --arch $a --steps 10 2>&1 | tail -1 | tee -a gpurun_out/net_${TAG}.txt
done
python tools/generic_bench.py --iters 5 > gpurun_out/generic_${TAG}.txt 2>&1
# round 5: the GROWN network (raw width multiplier 1.5: 78 / 156 / 313 / 627 channels) and the 3-task sequence through CPGSession
python bench.py --width-multiplier 1.5 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_${TAG}_grown.log 2>&1; tail -1 gpurun_out/bench_${TAG}_grown.log | cut -c1-200
python tools/conv_bench.py --width-multiplier 1.5 --iters 5 > gpurun_out/conv_bench_${TAG}_grown.txt 2>&1; tail -3 gpurun_out/conv_bench_${TAG}_grown.txt
python bench.py --task-sequence 3 > gpurun_out/bench_${TAG}_seq3.log 2>&1; tail -1 gpurun_out/bench_${TAG}_seq3.log | cut -c1-200
# round 6: configs[3] / configs[4] as sequences, the 6-task VGG16 sequence (growth at task 4), the 220-step cycles of the other two topologies
for a in resnet50 spherenet20; do
  python bench.py --task-sequence 3 --arch $a > gpurun_out/bench_${TAG}_seq3_$a.log 2>&1; tail -1 gpurun_out/bench_${TAG}_seq3_$a.log | cut -c1-200
  python bench.py --arch $a --no-cpu-baseline > gpurun_out/bench_${TAG}_${a}_k220.log 2>&1; tail -1 gpurun_out/bench_${TAG}_${a}_k220.log | cut -c1-200
done
python bench.py --task-sequence 6 --grow-at-task 4 --sequence-sparsity 0.5 > gpurun_out/bench_${TAG}_seq6.log 2>&1; tail -1 gpurun_out/bench_${TAG}_seq6.log | cut -c1-200
# interleaved A/B of the launch diet (ABI 3): default | self-packing conv calls | per-layer optimizer launches | both off -- three rounds, one box
: > gpurun_out/ab_launch_diet_${TAG}.txt
for a in resnet50 spherenet20; do for round in 1 2 3; do
  for v in "CPG_X=1" "CPG_PACK_CACHE=0" "CPG_MULTI_TENSOR=0" "CPG_PACK_CACHE=0 CPG_MULTI_TENSOR=0"; do
    echo -n "$a round $round  $v  " >> gpurun_out/ab_launch_diet_${TAG}.txt
    env $v python bench.py --arch $a --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-clock 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms/step')" >> gpurun_out/ab_launch_diet_${TAG}.txt
  done
done; done
cat gpurun_out/ab_launch_diet_${TAG}.txt
python tools/conv_bench.py --iters 5 > gpurun_out/conv_bench_${TAG}.txt 2>&1; tail -3 gpurun_out/conv_bench_${TAG}.txt
cd /tmp && export TMPDIR=/tmp
for cfg in "vgg16:--arch vgg16" "resnet50:--arch resnet50" "spherenet20:--arch spherenet20" "task2:--task 2" "b32:--batch 32" "grown:--width-multiplier 1.5"; do
  a=${cfg%%:*}; flags=${cfg#*:}
  rm -rf $R/gpurun_out/prof_${TAG}_$a
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_$a -o run -- python $R/bench.py $flags --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --optin-steps 0 > $R/gpurun_out/prof_${TAG}_$a.log 2>&1
  db=$(find $R/gpurun_out/prof_${TAG}_$a -name '*.db' | head -1)
  python $R/tools/rocprof_summary.py $db 60 > $R/gpurun_out/summary_${TAG}_$a.md 2>&1
  rm -rf $R/gpurun_out/prof_${TAG}_$a
done
# HBM traffic of the bench's own launch mix: FETCH_SIZE and WRITE_SIZE in separate counter passes (kernel trace only, every launch clocked; a pass
# takes ~10 s -- bounded at 300 s and retried once: rocprofv3 has hung in its signal handler on this pool)
cd $R
COMMIT=${COMMIT:-unknown} LIMIT=300 bash tools/run_traffic.sh
for a in vgg16 resnet50 spherenet20; do for c in FETCH_SIZE WRITE_SIZE; do
  if [ ! -e gpurun_out/btraffic/${a}_$c/.collected ]; then ARCHS=$a COMMIT=${COMMIT:-unknown} LIMIT=300 bash tools/run_traffic.sh; break; fi
done; done
cp gpurun_out/traffic.json gpurun_out/traffic_${TAG}.json
rm -rf gpurun_out/btraffic
