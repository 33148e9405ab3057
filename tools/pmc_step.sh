# One SQ counter pass over two train steps of each topology -> gpurun_out/pmc_step_<arch>.md (tools/pmc_step_table.py)
R=$PWD
cd /tmp && export TMPDIR=/tmp
for a in ${ARCHS:-vgg16 resnet50 spherenet20}; do
  rm -rf $R/gpurun_out/pmc_step/$a
  timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY --kernel-trace -d $R/gpurun_out/pmc_step/$a -o run --output-format csv -- python $R/tools/net_bench.py --arch $a --steps 1 > /dev/null 2>&1
  python $R/tools/pmc_step_table.py $R/gpurun_out/pmc_step/$a > $R/gpurun_out/pmc_step_$a.md 2>&1
done
rm -rf $R/gpurun_out/pmc_step
