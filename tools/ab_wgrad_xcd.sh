R=$GRAFT_REPO_ROOT
cd $R
for v in 0 1; do echo "CPG_WW_XCD=$v"; CPG_WW_XCD=$v python tools/conv_bench.py --iters 10 --only wgrad 2>&1 | tail -1; done
for v in 0 1; do echo "CPG_WW_XCD=$v"; CPG_WW_XCD=$v python tools/conv_bench.py --iters 10 --only wgrad 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/tr_$v_$c
    CPG_WW_XCD=$v rocprofv3 --pmc $c --kernel-trace -d /tmp/tr_${v}_$c -o run --output-format csv -- python $R/tools/conv_bench.py --pmc-pass --only wgrad > /dev/null 2>&1
    python - <<P
import csv, glob, collections
f=glob.glob('/tmp/tr_${v}_$c/**/*counter_collection.csv', recursive=True)[0]
acc=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0][:40]
    acc[k]+=float(r['Counter_Value']); n[k]+=1
for k,v in sorted(acc.items(), key=lambda kv:-kv[1])[:4]:
    print('XCD=$v $c', k, n[k], '%.1f MB (raw counter units x 1e-6)' % (v/1e6))
P
  done
done
