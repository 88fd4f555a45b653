import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "traffic-env_amd")]
import numpy as np, torch
from gym_traffic.core import TfxEngine
from gym_traffic import workload as wl
def t(eng, n=200, reps=6):
    eng.step(n); torch.cuda.synchronize()
    out = []
    for i in range(reps):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(); eng.step(n); ev1.record(); torch.cuda.synchronize()
        out.append(ev0.elapsed_time(ev1) * 1e3 / n)
    return min(out)
for E in (16, 1024):
    eng = TfxEngine(4, 4, 200.0, 34, n_envs=E, planes=2)
    eng.reset(np.zeros((1, eng.I), np.int32)); eng.set_spawns(); eng.set_actions(cycle_period=20)
    print("cfg1 shape x %d, EMPTY roads, cycle lights: %.2f us/tick" % (E, t(eng)))
    eng2 = TfxEngine(4, 4, 200.0, 34, n_envs=E, planes=2)
    eng2.reset(np.zeros((1, eng2.I), np.int32)); eng2.set_spawns(period=8); eng2.set_actions(cycle_period=20)
    print("cfg1 shape x %d, empty start + periodic spawns: %.2f us/tick (cars/road now %.1f)" % (E, t(eng2), float(eng2.cars_on_roads_flat().float().mean())))
    eng3 = wl.setup_engine("cfg1", envs=E)
    print("cfg1 x %d prefilled: %.2f us/tick (cars/road %.1f)" % (E, t(eng3), float(eng3.cars_on_roads_flat().float().mean())))
