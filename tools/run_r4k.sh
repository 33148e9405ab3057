R=$PWD
cd /tmp && export TMPDIR=/tmp
for cfg in "r50t1:--arch resnet50" "r50t2:--arch resnet50 --task 2"; do
  a=${cfg%%:*}; flags=${cfg#*:}
  rm -rf $R/gpurun_out/prof_$a
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$a -o run -- python $R/bench.py $flags --steps 20 --warmup 5 --no-cpu-baseline --optin-steps 0 > $R/gpurun_out/prof_$a.log 2>&1
  db=$(find $R/gpurun_out/prof_$a -name '*.db' | head -1)
  python $R/tools/rocprof_summary.py $db 70 > $R/gpurun_out/summary_$a.md 2>&1
  rm -rf $R/gpurun_out/prof_$a
done
