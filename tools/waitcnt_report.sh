#!/bin/bash
# s_waitcnt vmcnt(N) histogram of one kernel of a variant build: tools/waitcnt_report.sh "<-D flags>" [mangled kernel name]
cd "$(dirname "$0")/.."
K=${2:-_ZN3tfx9k_move_ttILb1ELb0ELb0EEEvNS_3DevEii}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-gpu-flush-denormals-to-zero -fno-fast-math -Iinclude $1 --cuda-device-only -S -o /tmp/wc.s traffic-env_amd/csrc/tfx_hip.hip 2>/dev/null
python3 - "$K" <<'PY'
import re, sys, collections
s = open('/tmp/wc.s').read(); name = sys.argv[1]
a = s.index(name + ':'); b = s.index('.Lfunc_end', a); body = s[a:b]
c = collections.Counter(re.findall(r's_waitcnt vmcnt\(\d+\)', body))
print(dict(c), len(re.findall(r'global_load_dwordx2', body)), 'x2 loads', len(re.findall(r'global_store_dwordx2', body)), 'x2 stores',
      re.search(r'\.vgpr_count:\s+(\d+)', s[s.index('.name:           ' + name):]).group(0) if ('.name:           ' + name) in s else '')
PY
