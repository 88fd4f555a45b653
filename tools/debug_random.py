import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'traffic-env_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch
from test_gpu_parity import *
from oracle.oracle import ring_order

m, n, C, length, validate, sorted_x = 2, 2, 10, 60.0, False, True
rng = np.random.RandomState(1234 + C + int(sorted_x))
E = 6
eng = engine_for(dict(m=m, n=n, length=length, capacity=C, rate=0.5, validate=validate), n_envs=E)
orc = oracle_like(eng)
x, v, w, leading, lastcar = random_state(rng, E, eng.R, C, length, crowd=rng.choice([0.3, 0.8]), beyond=rng.choice([0.0, 0.05, 0.4]), sorted_x=sorted_x)
phase = rng.randint(2, size=(E, eng.I)).astype(np.int32)
elapsed = rng.randint(0, 12, size=(E, eng.I)).astype(np.int32)
load_both(eng, orc, x, v, w, leading, lastcar, phase, elapsed)
eng.set_tick(60); orc.steps[:] = 60
act = rng.randint(2, size=(E, eng.I)).astype(np.int32)
roads = [rng.choice(eng.entrypoints, size=rng.randint(0, 4)).tolist() for _ in range(E)]
eng.set_spawns(counts=counts(eng, roads)); eng.set_actions(act); eng.step(1)
orc.step(act, roads)
ld, lc = eng.leading.cpu().numpy(), eng.lastcar.cpu().numpy()
st = eng.state.cpu().numpy()
print('nexts', eng.nexts.tolist())
for k in range(E):
    for e in range(eng.R):
        if ld[k,e] != orc.leading[k,e] or lc[k,e] != orc.lastcar[k,e]:
            print('IDX env', k, 'road', e, 'gpu', ld[k,e], lc[k,e], 'orc', orc.leading[k,e], orc.lastcar[k,e], 'init', leading[k,e], lastcar[k,e]); continue
        for s in ring_order(int(ld[k,e]), int(lc[k,e]), C):
            for p, nm, o in ((0,'x',orc.x),(1,'v',orc.v),(2,'w',orc.w)):
                if st[k,e,p,s] != o[k,e,s] and not (np.isnan(st[k,e,p,s]) and np.isnan(o[k,e,s])):
                    print('VAL env', k, 'road', e, 'slot', s, nm, 'gpu', st[k,e,p,s], 'orc', o[k,e,s], 'init ld/lc', leading[k,e], lastcar[k,e], 'now', ld[k,e], lc[k,e], 'pred', [q for q in range(eng.R) if eng.nexts[q]==e], 'spawn', roads[k])
