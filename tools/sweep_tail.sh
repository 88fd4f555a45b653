#!/bin/bash
# k_tail workgroup size / residency sweep at cfg2 (run through gpurun): bash tools/sweep_tail.sh <outdir under gpurun_out>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-sweep_tail}; mkdir -p $O; cd $R
run() { # label, env...
  local label=$1; shift
  env "$@" python3 bench.py --steps 100 --warmup 20 --repeats 3 --no-cpu-baseline > $O/$label.json 2> $O/$label.err
  python3 - "$O/$label.json" "$label" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-28s ms/tick %.4f  spread %.4f-%.4f  agent %.3f ms  k_tail alone %.4f" % (sys.argv[2], d["ms_per_step"], d["ms_per_step_spread"][0], d["ms_per_step_spread"][1], d["agent_decision_ms"], d["roofline"]["rest_of_tick_ms"]))
PY
}
run base X=1
for t in 64 192 256; do run half$t TFX_TAIL_THREADS_HALF=$t; done
for b in 2 3 6; do run half128_b$b TFX_TAIL_BLOCKS_PER_CU=$b; done
run half256_b2 TFX_TAIL_THREADS_HALF=256 TFX_TAIL_BLOCKS_PER_CU=2
run half256_b3 TFX_TAIL_THREADS_HALF=256 TFX_TAIL_BLOCKS_PER_CU=3
