#!/bin/bash
# Times bench.py (cfg2, env halves serialised so that launch_ms is the pass's own) for every experiment library
# under traffic-env_amd/lib/exp (make -C traffic-env_amd/csrc exp EXP="name:-Dflag ...").  usage: tools/exp_variants.sh [names...]
cd ${GRAFT_REPO_ROOT:-.}
names=${@:-$(ls traffic-env_amd/lib/exp | sed 's/libtfx_\(.*\)\.so/\1/')}
for n in $names; do
  for blocks in ${BLOCKS:-0}; do
  TFX_LIB=$PWD/traffic-env_amd/lib/exp/libtfx_$n.so TFX_MOVE_BLOCKS_PER_CU=$blocks python3 bench.py --no-cpu-baseline --steps ${STEPS:-100} 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('%-10s blocks/CU %s: value %.4g  ms/tick %.4f  pass %.4f ms  rest/tick %.4f  frac %.3f' % ('$n','$blocks',d['value'],d['ms_per_step'],r['launch_ms'],r['rest_of_tick_ms'],r['frac']))"
  done
done
