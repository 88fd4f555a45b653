"""cfg4 (BASELINE.json's stress configuration): 64x64 grid, 128-car roads (CAPACITY = 130), empty
start, on-device Poisson arrivals (local_cars_per_sec = 0.12 -> 15.36 cars/tick/env) and the greedy
controller every 3 ticks, 2000 ticks with zero host involvement.  Prints the occupancy histogram
(divergent roads), overflow count and throughput."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "traffic-env_amd")]
import numpy as np, torch
from gym_traffic.core import TfxEngine

def run(E, ticks=2000, lcps=0.12):
    m = n = 64
    eng = TfxEngine(m, n, 800.0, 130, n_envs=E, planes=2)
    eng.reset(np.zeros((1, eng.I), np.int32))
    eng.set_poisson(lcps * m * 4 * 0.5, seed=1234)
    eng.set_greedy(3)
    eng.step(50); torch.cuda.synchronize(); eng.reset_counters()
    t0 = time.perf_counter()
    overflow_ticks = 0
    for _ in range(ticks // 50):
        eng.step(50)
        overflow_ticks += int(eng.done.sum())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    upd = eng.vehicle_updates()
    occ = eng.cars_on_roads_flat().cpu().numpy().ravel()
    hist = np.bincount(np.minimum(occ // 16, 8), minlength=9)
    print("cfg4 E=%d: %d ticks in %.2f s = %.0f env-ticks/s, %.3e vehicle-updates/s, %d cars on the roads, "
          "max/road %d, 50-tick windows with an overflow (summed over envs): %d"
          % (E, ticks, dt, E * ticks / dt, upd / dt, occ.sum(), occ.max(), overflow_ticks))
    print("  roads by occupancy [0-15,16-31,...,112-127,128]: %s" % hist.tolist())

for E in (1, 16):
    run(E)
