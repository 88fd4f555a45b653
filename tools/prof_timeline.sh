#!/bin/bash
# kernel start/end timeline of the default (split) bench run: how the two env halves' launches overlap
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/timeline; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/kt -o p --output-format csv -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $O/kt.log 2>&1
cd $R; python3 - <<'PY'
import csv,glob,os
O=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/timeline"
rows=[]
for f in glob.glob(O+"/kt/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "k_move_tt" in n or "k_tail" in n:
            rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"pass" if "move_tt" in n else "tail",r.get("Queue_Id","?"),r.get("Stream_Id","?")))
rows.sort()
t0=rows[0][0]
# the timed region is the first 40 ticks after warmup: print launches 20..60
for s,e,k,q,st in rows[20:60]:
    print("%9.1f %9.1f %6.1f us  %s  q%s s%s" % ((s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3,k,q,st))
PY
