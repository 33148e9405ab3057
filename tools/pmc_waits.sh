# Where the non-MFMA, non-VALU cycles of the Winograd kernels go: wait / LDS / vector-memory counters on one VGG16 layer
#   LAYER=f27 bash tools/pmc_waits.sh -> gpurun_out/pmc_waits.txt
R=$PWD
L=${LAYER:-f27}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_waits
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU --kernel-trace -d $R/gpurun_out/pmc_waits/a -o run --output-format csv -- python $R/tools/conv_bench.py --layers $L --iters 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/pmc_waits/b -o run --output-format csv -- python $R/tools/conv_bench.py --layers $L --iters 1 > /dev/null 2>&1
python - <<PY > $R/gpurun_out/pmc_waits.txt 2>&1
import collections, csv, glob
for d in ('a', 'b'):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('$R/gpurun_out/pmc_waits/%s/**/*counter_collection.csv' % d, recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            per[(r['Kernel_Name'], r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
        for (k, _), c in per.items():
            for n, v in c.items():
                agg[k][n].append(v)
    for k, c in agg.items():
        if not any(t in k for t in ('k_wg3', 'k_wgw', 'k_wg1<')):
            continue
        m = {n: sum(v) / len(v) for n, v in c.items()}
        cyc = m.get('SQ_BUSY_CYCLES', 0) / 32
        print(d, k[:90], 'cycles %.3e' % cyc)
        for n, v in sorted(m.items()):
            print('   %-26s %.4e   per SIMD-cycle %.3f' % (n, v, v / 1024 / cyc if cyc else 0))
PY
rm -rf $R/gpurun_out/pmc_waits
