#!/bin/bash
# experiment libraries (make exp) at cfg2: bash tools/sweep_tail2.sh <outdir> <lib names...>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-sweep}; shift; mkdir -p $O; cd $R
for n in "$@"; do
  lib=$R/traffic-env_amd/lib/exp/libtfx_$n.so
  [ "$n" = base ] && lib=$R/traffic-env_amd/lib/libtfx_hip.so
  TFX_LIB=$lib python3 bench.py --steps 100 --warmup 20 --repeats 3 --no-cpu-baseline > $O/$n.json 2> $O/$n.err
  python3 - "$O/$n.json" "$n" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-20s ms/tick %.4f  (%.4f-%.4f)  agent %.3f ms  pass %.4f  k_tail alone %.4f" % (sys.argv[2], d["ms_per_step"], d["ms_per_step_spread"][0], d["ms_per_step_spread"][1], d["agent_decision_ms"], d["roofline"]["launch_ms"], d["roofline"]["rest_of_tick_ms"]))
PY
done
