set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --kernel-trace --pmc $c -d $O/$c -o p --output-format csv -- python3 $R/bench.py --steps 20 --warmup 4 --no-cpu-baseline > $O/$c.log 2>&1
done
rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python3 $R/bench.py --steps 20 --warmup 4 --no-cpu-baseline > $O/kt.log 2>&1
cd $R; python3 - <<'PY'
import csv,glob,collections,os
O=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/r3c"
for c in ("FETCH_SIZE","WRITE_SIZE"):
    agg=collections.defaultdict(list)
    for f in glob.glob(O+"/"+c+"/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(c,k,len(v),sum(v)/len(v))
for f in glob.glob(O+"/kt/**/*kernel_stats.csv",recursive=True):
    for r in csv.DictReader(open(f)): print(r["Name"][:60],r["Calls"],r["AverageNs"])
PY
