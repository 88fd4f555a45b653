#!/bin/bash
# rocprofv3 kernel trace of fused agent decisions next to plain 10-tick calls at cfg2 (run through gpurun):
#   bash tools/prof_agent.sh <outdir under gpurun_out>
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-prof_agent}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/agent -o p --output-format csv -- python3 $R/tools/bench_agent_step.py > $O/agent.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/plain -o p --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --call-ticks 10 --repeats 2 --no-cpu-baseline > $O/plain.log 2>&1
cd $R
for k in agent plain; do echo "== $k"; f=$(find $O/$k -name '*kernel_stats.csv' | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-70s calls %6s  avg %10.1f us  total %10.1f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
tail -2 $O/agent.log
