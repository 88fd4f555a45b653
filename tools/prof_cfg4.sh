set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3g; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python3 $R/tools/run_cfg4.py > $O/kt.log 2>&1
cd $R; python3 - <<'PY'
import csv,glob,os
O=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/r3g"
for f in glob.glob(O+"/kt/**/*kernel_stats.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if 'tfx' in r["Name"]: print(r["Name"][:70],r["Calls"],r["AverageNs"], r["TotalDurationNs"])
PY
tail -4 $O/kt.log
