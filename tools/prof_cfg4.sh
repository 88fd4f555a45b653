#!/bin/bash
# rocprofv3 --kernel-trace --stats of the cfg4 closed loop at E = 1 and E = 16 (run through gpurun):
#   bash tools/prof_cfg4.sh <outdir under gpurun_out>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-prof_cfg4}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for E in 1 16; do
  python3 $R/tools/c4_loop.py $E > $O/loop_E$E.txt 2>&1
  rocprofv3 --kernel-trace --stats -d $O/E$E -o p --output-format csv -- python3 $R/tools/c4_loop.py $E 10 > $O/E$E.log 2>&1
  f=$(find $O/E$E -name '*kernel_stats.csv' | head -1)
  echo "== E=$E: $(tail -1 $O/loop_E$E.txt)"
  python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("  %-64s calls %6s  avg %9.2f us  %5s %%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done
