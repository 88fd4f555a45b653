"""Throughput of validate mode (the reference's `--mode validate`: every car carries its spawn tick, cars leaving the map
log their trip time, advance_hack traffic_env.py:139-157) at the headline shape: 4096 envs of 16x16 x 64-car roads, the
benchmark's prefill, periodic arrivals, fixed-cycle lights.  24 B per car and tick-pass instead of 16."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "traffic-env_amd")]
import numpy as np, torch
from gym_traffic import workload as wl
from gym_traffic.core import TfxEngine

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
c = wl.CONFIGS["cfg2"]
eng = TfxEngine(c["m"], c["n"], c["length"], c["capacity"], n_envs=E, planes=3, validate=True, trip_cap=64)
eng.reset(np.zeros((1, eng.I), np.int32))
x, v, ld, lc = wl.prefill_one_env(c["m"], c["n"], c["length"], c["capacity"], c["prefill"], c["gap"])
ring = eng.xv
ring[..., 0].copy_(torch.as_tensor(x).to(eng.device)[None].expand_as(ring[..., 0]))
ring[..., 1].copy_(torch.as_tensor(v).to(eng.device)[None].expand_as(ring[..., 1]))
eng.leading[:] = torch.as_tensor(ld).to(eng.device)[None]
eng.lastcar[:] = torch.as_tensor(lc).to(eng.device)[None]
eng.refresh(); eng.drop_staging()
eng.set_spawns(period=wl.SPAWN_PERIOD); eng.set_actions(cycle_period=wl.LIGHT_PERIOD)
eng.step(100); torch.cuda.synchronize(); eng.reset_counters(); p0 = eng.pair_ticks()
t0 = time.perf_counter(); eng.step(200); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("validate mode, %d envs of cfg2: %.4f ms per tick, %.3e vehicle-updates/s (%s, %d of 200 ticks in pairs, %d trips logged)"
      % (E, dt / 200 * 1e3, eng.vehicle_updates() / dt, eng.step_kernel(), eng.pair_ticks() - p0, int(eng.n_trips.sum())))
