#!/bin/bash
# The FETCH_SIZE correction, measured on a stream of KNOWN size: tools/copy_probe.hip reads (and writes) 1 835 008 000
# bytes per k_rw launch, 4, 8 or 16 bytes per lane.  Prints FETCH_SIZE / WRITE_SIZE (KiB) per launch and their ratio to
# the bytes really read / written.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fetch_factor; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $O/copy_probe $R/tools/copy_probe.hip || exit 1
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/$c -o p --output-format csv -- $O/copy_probe > $O/$c.log 2>&1
done
cd $R; python3 - <<'PY'
import csv, glob, os, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/fetch_factor"
BYTES = 1835008000.0
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob(O + "/" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_rw" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:72]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        m = sum(v) / len(v)
        print("%-10s %-74s %3d launches  %12.0f KiB  = %.3f x the %.0f bytes each launch %s" %
              (c, k, len(v), m, m * 1024 / BYTES, BYTES, "reads" if c == "FETCH_SIZE" else "writes"))
PY
