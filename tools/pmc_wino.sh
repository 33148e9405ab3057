# SQ-only counter passes over the Winograd kernels (TA_* counters hung the profiler on this pool: do not add them)
#   LAYER=f27 bash tools/pmc_wino.sh   -> gpurun_out/pmc_w1 (forward / input gradient), gpurun_out/pmc_ww (weight gradient)
R=$PWD
L=${LAYER:-f27}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_w1 $R/gpurun_out/pmc_ww
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace -d $R/gpurun_out/pmc_w1 -o run --output-format csv -- python $R/tools/wino_bench.py --layers $L --iters 1 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_ww -o run --output-format csv -- python $R/tools/wino_wgrad_bench.py --layers $L --iters 1 > /dev/null 2>&1
ls $R/gpurun_out/pmc_w1 $R/gpurun_out/pmc_ww
