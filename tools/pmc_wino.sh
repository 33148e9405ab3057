R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_w1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace -d $R/gpurun_out/pmc_w1 -o run --output-format csv -- python $R/tools/wino_bench.py --layers f27 --iters 1 > /dev/null 2>&1
ls $R/gpurun_out/pmc_w1
