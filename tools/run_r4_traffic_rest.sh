R=$PWD
cd /tmp && export TMPDIR=/tmp
for a in ${ARCHS:-resnet50 spherenet20}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 600 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/btraffic/${a}_$c -o run --output-format csv -- python $R/bench.py --arch $a --steps 20 --warmup 1 --no-cpu-baseline --optin-steps 0 --clock-every 1 > $R/gpurun_out/btraffic_${a}_$c.log 2>&1
    ls -la $R/gpurun_out/btraffic/${a}_$c | tail -3
  done
done
