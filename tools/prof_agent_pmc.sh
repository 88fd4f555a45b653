#!/bin/bash
# HBM bytes of the kernels of fused agent decisions at cfg2 (FETCH_SIZE and WRITE_SIZE in passes of their own):
#   bash tools/prof_agent_pmc.sh <outdir under gpurun_out>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-prof_agent_pmc}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/$c -o p --output-format csv -- python3 $R/tools/bench_agent_step.py > $O/$c.log 2>&1 || echo "FAILED $c"
done
cd $R
python3 - "$O" <<'PY'
import csv, sys, collections
O = sys.argv[1]
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
for i, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    import glob
    f = glob.glob(O + "/" + c + "/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        n = r["Kernel_Name"].replace("void tfx::", "").replace("tfx::", "").split("(")[0]
        tot[n][i] += float(r["Counter_Value"])
        if i == 0: tot[n][2] += 1
for n, (fe, wr, k) in sorted(tot.items(), key=lambda x: -(2 * x[1][0] + x[1][1])):
    if k: print("%-60s launches %5d  read %8.1f MB  written %8.1f MB  per launch" % (n[:60], k, 2 * fe * 1024 / k / 1e6, wr * 1024 / k / 1e6))
PY
