"""Throughput of the heterogeneous-car path (archetype table, tfx_config.n_archetypes) at the headline shape: 4096 envs of
16x16 x 64-car roads, the benchmark's prefill with every car given one of three table rows, periodic arrivals (row 0),
fixed-cycle lights.  The HET forms of the two-tick pass and k_tail (TFX_PAIRS=0: k_move_t<HET> + k_advance, tick by
tick); 24 B per car and pass."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "traffic-env_amd")]
import numpy as np, torch
from gym_traffic import workload as wl
from gym_traffic.core import TfxEngine

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
c = wl.CONFIGS["cfg2"]
tab = [[11.11, 4, 3, 4, 13.89, 6, 2, 1], [8.0, 8, 1.5, 4, 10.0, 4, 2.5, 2], [12.0, 3.5, 4, 2, 16.0, 7, 1.5, 1]]


def make_engine():
    eng = TfxEngine(c["m"], c["n"], c["length"], c["capacity"], n_envs=E, planes=3, archetypes=tab)
    eng.reset(np.zeros((1, eng.I), np.int32))
    x, v, ld, lc = wl.prefill_one_env(c["m"], c["n"], c["length"], c["capacity"], c["prefill"], c["gap"])
    ring = eng.xv
    ring[..., 0].copy_(torch.as_tensor(x).to(eng.device)[None].expand_as(ring[..., 0]))
    ring[..., 1].copy_(torch.as_tensor(v).to(eng.device)[None].expand_as(ring[..., 1]))
    eng._ringa.copy_(torch.randint(0, 3, eng._ringa.shape, dtype=torch.uint8, device=eng.device))
    eng.leading[:] = torch.as_tensor(ld).to(eng.device)[None]
    eng.lastcar[:] = torch.as_tensor(lc).to(eng.device)[None]
    eng.refresh(); eng.drop_staging()
    eng.set_spawns(period=wl.SPAWN_PERIOD); eng.set_actions(cycle_period=wl.LIGHT_PERIOD)
    return eng


eng = make_engine()
eng.step(100); torch.cuda.synchronize(); eng.reset_counters()
t0 = time.perf_counter(); eng.step(200); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("heterogeneous cars, %d envs of cfg2, 3 archetypes: %.4f ms per tick, %.3e vehicle-updates/s (%s)"
      % (E, dt / 200 * 1e3, eng.vehicle_updates() / dt, eng.step_kernel()))
del eng
# fused 10-tick decisions (tfx_agent_step) of the same batch, early in the run (before the slower rows' queues reach
# the ring capacity: an env that overflows stands still for the rest of its decision)
eng = make_engine()
eng.step(10)
eng.agent_step(10, remi=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    adone = eng.agent_step(10, remi=True)[2]
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print("  10-tick agent decision: %.3f ms (%.4f ms per tick), %d envs done in the last one" % (dt * 1e3, dt * 1e2, int(adone.sum())))
