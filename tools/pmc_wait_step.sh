# Per kernel of a train step: share of the wave cycles spent in s_waitcnt (SQ_WAIT_ANY), waiting on LDS, with a vector / LDS instruction active
R=$PWD
cd /tmp && export TMPDIR=/tmp
for a in ${ARCHS:-vgg16 resnet50 spherenet20}; do
  rm -rf $R/gpurun_out/pmc_wait/$a
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/pmc_wait/$a -o run --output-format csv -- python $R/tools/net_bench.py --arch $a --steps 1 > /dev/null 2>&1
  python - <<PY > $R/gpurun_out/pmc_wait_$a.md 2>&1
import collections, csv, glob, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob('$R/gpurun_out/pmc_wait/$a/**/*counter_collection.csv', recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        k = re.sub(r'^void ', '', k).split('(')[0][:100]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if (k, r['Dispatch_Id']) not in seen:
            seen.add((k, r['Dispatch_Id'])); calls[k] += 1
tot = sum(c['SQ_BUSY_CYCLES'] for c in agg.values())
print('| kernel | calls | share of cycles | MFMA busy | waiting in s_waitcnt | ... on LDS | vector instruction active | LDS instruction active |')
print('|---|---:|---:|---:|---:|---:|---:|---:|')
for k, c in sorted(agg.items(), key=lambda kv: -kv[1]['SQ_BUSY_CYCLES']):
    if c['SQ_BUSY_CYCLES'] / tot < 0.004 or not c['SQ_WAVE_CYCLES']:
        continue
    w = c['SQ_WAVE_CYCLES']
    cyc = c['SQ_BUSY_CYCLES'] / 32
    print('| \`%s\` | %d | %.3f | %.3f | %.3f | %.3f | %.3f | %.3f |' % (k, calls[k], c['SQ_BUSY_CYCLES'] / tot, c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc, c['SQ_WAIT_ANY'] / w, c['SQ_WAIT_INST_LDS'] / w, c['SQ_ACTIVE_INST_VALU'] / w, c['SQ_ACTIVE_INST_LDS'] / w))
PY
done
rm -rf $R/gpurun_out/pmc_wait
