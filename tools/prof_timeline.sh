#!/bin/bash
# kernel start/end timeline of a split run: how the two env halves' launches overlap.
#   tools/prof_timeline.sh step   (bench.py, tfx_step)      tools/prof_timeline.sh agent   (tools/bench_agent_step.py)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/timeline_${1:-step}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
if [ "${1:-step}" = agent ]; then
  rocprofv3 --kernel-trace -d $O/kt -o p --output-format csv -- python3 $R/tools/bench_agent_step.py cfg2 > $O/kt.log 2>&1
else
  rocprofv3 --kernel-trace -d $O/kt -o p --output-format csv -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $O/kt.log 2>&1
fi
cd $R; O=$O python3 - <<'PY'
import csv,glob,os
O=os.environ["O"]
rows=[]
for f in glob.glob(O+"/kt/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        for key,lab in (("k_move_tt","pass"),("k_tail","tail"),("k_risk","risk"),("k_advance","adv"),("k_edge","edge")):
            if key in n:
                rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),lab,r.get("Queue_Id","?")))
                break
rows.sort()
t0=rows[0][0]
for s,e,k,q in rows[60:110]:
    print("%9.1f %9.1f %6.1f us  %-5s q%s" % ((s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3,k,q))
PY
