# FETCH_SIZE / WRITE_SIZE passes over the linear layers at 32 rows (tools/linear_bench.py --batch 32, with and without a piggymask)
# -> gpurun_out/fc_small_traffic.md
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in plain pm; do
  flag=""; [ $v = pm ] && flag="--pm"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc_fc/$v/$c
    timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_fc/$v/$c -o run --output-format csv -- python $R/tools/linear_bench.py --batch 32 --iters 3 $flag > $R/gpurun_out/pmc_fc_${v}_$c.log 2>&1
  done
done
{ for v in plain pm; do echo "## linear_bench.py --batch 32 $([ $v = pm ] && echo --pm)"; echo; python $R/tools/pmc_kernels.py $R/gpurun_out/pmc_fc/$v 'k_fc|k_pw|k_gemm|k_split|k_bias'; echo; done; } > $R/gpurun_out/fc_small_traffic.md 2>&1
rm -rf $R/gpurun_out/pmc_fc
cat $R/gpurun_out/fc_small_traffic.md
