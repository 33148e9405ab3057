# SQ counter pass over the pointwise (1x1) kernels of three ResNet-50 shapes -> gpurun_out/pmc_pw.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_pw
for s in "256>1024 @14" "256>128 @56" "64>256 @56"; do
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace -d "$R/gpurun_out/pmc_pw/a_${s// /_}" -o run --output-format csv -- python $R/tools/generic_bench.py --iters 1 --only "$s" > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d "$R/gpurun_out/pmc_pw/b_${s// /_}" -o run --output-format csv -- python $R/tools/generic_bench.py --iters 1 --only "$s" > /dev/null 2>&1
done
for d in $R/gpurun_out/pmc_pw/*; do echo "=== $d"; python $R/tools/pmc_wino_table.py "$d"; done > $R/gpurun_out/pmc_pw.txt 2>&1
rm -rf $R/gpurun_out/pmc_pw
