#!/bin/bash
# bench.py (cfg2) for several values of TFX_MOVE_BLOCKS_PER_CU: tools/sweep_blocks.sh 6 8 10 12
cd ${GRAFT_REPO_ROOT:-.}
for b in "$@"; do
  TFX_MOVE_BLOCKS_PER_CU=$b python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('blocks/CU %s: %.4g  %.4f ms/tick  pass %.4f ms' % ('$b', d['value'], d['ms_per_step'], d['roofline']['launch_ms']))"
done
