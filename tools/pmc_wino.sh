# SQ-only counter passes over the Winograd kernel (TA_* counters hung the profiler on this pool: do not add them)
R=$PWD
L=${LAYER:-f17}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_wg*
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --kernel-trace -d $R/gpurun_out/pmc_wgA -o run --output-format csv -- python $R/tools/wino_bench.py --layers $L --iters 1 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --kernel-trace -d $R/gpurun_out/pmc_wgC -o run --output-format csv -- python $R/tools/wino_bench.py --layers $L --iters 1 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_ACTIVE_INST_MISC --kernel-trace -d $R/gpurun_out/pmc_wgD -o run --output-format csv -- python $R/tools/wino_bench.py --layers $L --iters 1 > /dev/null 2>&1
ls $R/gpurun_out/pmc_wgA $R/gpurun_out/pmc_wgC $R/gpurun_out/pmc_wgD
