# FETCH_SIZE / WRITE_SIZE passes over bench.py itself for the topologies in ARCHS (separate counter passes, kernel trace only, every launch clocked,
# bounded): gpurun_out/btraffic/<arch>_<counter>/  ->  python tools/bench_traffic.py gpurun_out/btraffic <commit> > profiles/r05_traffic_bench.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
for a in ${ARCHS:-vgg16 resnet50 spherenet20}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/btraffic/${a}_$c
    timeout -s KILL ${LIMIT:-1200} rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/btraffic/${a}_$c -o run --output-format csv -- python $R/bench.py --arch $a --steps 20 --warmup 1 --no-cpu-baseline --no-other-workloads --optin-steps 0 --clock-every 1 > $R/gpurun_out/btraffic_${a}_$c.log 2>&1
    [ -n "$(find $R/gpurun_out/btraffic/${a}_$c -name '*counter_collection.csv')" ] && touch $R/gpurun_out/btraffic/${a}_$c/.collected
    echo "$a $c: $(find $R/gpurun_out/btraffic/${a}_$c -name '*counter_collection.csv' | wc -l) csv, $(tail -c 300 $R/gpurun_out/btraffic_${a}_$c.log | tr '\n' ' ' | cut -c1-120)"
  done
done
python $R/tools/bench_traffic.py $R/gpurun_out/btraffic ${COMMIT:-unknown} > $R/gpurun_out/traffic.json 2> $R/gpurun_out/traffic.err
# keep only the counter CSVs small enough to travel back: the json is what is committed
find $R/gpurun_out/btraffic -name '*.csv' -size +20M -delete
