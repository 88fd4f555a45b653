"""Randomised differential test of the fused agent step (tfx_agent_step: Repeater + Remi on the device,
HIP graph replay) against the same wrappers emulated tick by tick on single-env oracles - random
grid, capacity, batch, layout, step path (k_res / per-tick kernels), decision length, periodic arrival density, with and without Remi,
including decisions cut short by an overflow.  FUZZ_SECS (default 300)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "traffic-env_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from gym_traffic.core import TfxEngine
from oracle.oracle import OracleEnv, live_mask
from test_gpu_agent_step import emulate_agent_step
from test_gpu_parity import random_state

rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t0, n, early = time.time(), 0, 0
while time.time() - t0 < float(os.environ.get("FUZZ_SECS", "300")):
    m, nn = int(rng.randint(1, 5)), int(rng.randint(1, 5))
    C = int(rng.choice([4, 6, 10, 14, 20, 34, 66]))
    L = float(rng.choice([40.0, 90.0, 150.0, 250.0]))
    E = int(rng.choice([1, 3, 7, 33]))
    layout = str(rng.choice(["ring", "transposed"]))
    remi = bool(rng.randint(2)); T = int(rng.choice([2, 3, 4, 5, 10])); period = int(rng.choice([1, 2, 5, 9]))
    os.environ["TFX_RESIDENT"] = str(int(rng.randint(3) > 0))     # the LDS-resident k_res (2 in 3) | per-tick kernels
    os.environ["TFX_RES_EPB"] = str(int(rng.choice([1, 2, 4])))
    os.environ["TFX_RES_LPR"] = str(int(rng.choice([1, 2, 3, 4])))
    os.environ["TFX_PAIRS"] = str(int(rng.choice([0, 2, 2])))     # two-tick passes + k_risk forced at any size | never
    os.environ["TFX_TAIL"] = str(int(rng.choice([0, 2])))         # (plain step() calls between decisions: k_tail, split)
    os.environ["TFX_SPLIT"] = str(int(rng.choice([0, 2])))
    os.environ["TFX_TT_SEG"] = str(int(rng.choice([0, 2])))        # the pass with every tile's walk split over wavefronts
    os.environ["TFX_TT_SEGS"] = str(int(rng.choice([2, 4, 8])))
    if rng.randint(4) == 0:                                        # one case in four: the handle's own choices
        for k in ("TFX_RESIDENT", "TFX_RES_EPB", "TFX_RES_LPR", "TFX_PAIRS", "TFX_TAIL", "TFX_SPLIT", "TFX_TT_SEG", "TFX_TT_SEGS"):
            os.environ.pop(k, None)
    val = bool(rng.randint(3) == 0)                               # validate mode: spawn ticks travel, trip times are logged
    # heterogeneous cars (one case in four on the transposed layout): a random table, the run starts from a random
    # ring state of mixed rows (arrivals are of row 0)
    het = layout == "transposed" and rng.randint(4) == 0
    tab8 = tab10 = None
    if het:
        na = int(rng.randint(1, 5))
        tab8 = np.stack([np.array([rng.uniform(5, 14), rng.choice([3.0, 4.0, 7.5, 12.0]), rng.uniform(0.8, 4), float(rng.randint(1, 9)),
                                   rng.uniform(8, 18), rng.uniform(2, 8), rng.uniform(1, 3), rng.uniform(0.5, 3)], np.float32) for _ in range(na)])
        if na == 1 and tab8[0, 3] == 4.0: tab8[0, 3] = 2.0
        tab10 = np.zeros((na, 10), np.float32); tab10[:, [1, 2, 3, 4, 5, 6, 7, 8]] = tab8
    eng = TfxEngine(m, nn, L, C, n_envs=E, planes=3 if (val or het or layout != "transposed") else 2, layout=layout, validate=val,
                    archetypes=tab8)
    orcs = [OracleEnv(m, nn, L, C, eng.dest, eng.phases, eng.nexts, validate=val) for _ in range(E)]
    ph = rng.randint(2, size=(E, eng.I)).astype(np.int32)
    eng.reset(ph)
    for k, o in enumerate(orcs):
        o.reset(ph[k])
    if het:
        x0, v0, w0, ld0, lc0 = random_state(rng, E, eng.R, C, L, crowd=rng.choice([0.2, 0.6]), beyond=rng.choice([0.0, 0.1]), sorted_x=True)
        arch0 = rng.randint(0, len(tab8), size=x0.shape).astype(np.uint8)
        eng.load_state(x0, v0, ld0, lc0, w=w0, arch=arch0)
        for k, o in enumerate(orcs):
            o.load_planes(0, x0[k], v0[k], w0[k], ld0[k], lc0[k], arch=arch0[k], archetypes=tab10)
        eng.set_tick(60)
        for o in orcs: o.steps[:] = 60
    eng.set_spawns(period=period)
    for step in range(int(rng.choice([4, 10, 16]))):
        act = rng.randint(2, size=(E, eng.I)).astype(np.int32)
        eng.set_actions(act)
        tick0 = eng.tick
        aobs, arew, adone = eng.agent_step(T, remi=remi)
        eobs, erew, edone = emulate_agent_step(orcs, tick0, act, eng.entrypoints, T, remi, period, archetypes=tab10)
        assert np.array_equal(adone.cpu().numpy(), edone), (n, step, "done")
        assert np.array_equal(aobs.cpu().numpy(), eobs), (n, step, "obs")
        assert np.array_equal(arew.cpu().numpy(), erew), (n, step, "reward")
        early += int(edone.sum())
        ld, lc = eng.leading.cpu().numpy(), eng.lastcar.cpu().numpy()
        xv = eng.xv.cpu().numpy()
        for k, o in enumerate(orcs):
            assert np.array_equal(ld[k], o.leading[0]) and np.array_equal(lc[k], o.lastcar[0]), (n, step, k)
            live = live_mask(ld[k], lc[k], C)
            assert np.array_equal(xv[k][live][:, 0].view(np.int32), o.x[0][live].view(np.int32)), (n, step, k)
            assert np.array_equal(xv[k][live][:, 1].view(np.int32), o.v[0][live].view(np.int32)), (n, step, k)
        if het:
            ar = eng.arch.cpu().numpy()
            for k, o in enumerate(orcs):
                lv = live_mask(ld[k], lc[k], C)
                assert np.array_equal(ar[k][lv], o.arch_plane(0, tab10)[lv]), (n, step, k, "rows")
        if val:
            w = eng.w.cpu().numpy()
            nt = eng.n_trips.cpu().numpy()
            for k, o in enumerate(orcs):
                assert np.array_equal(w[k][live_mask(ld[k], lc[k], C)], o.w[0][live_mask(ld[k], lc[k], C)]), (n, step, k, "w")
                assert nt[k] == o.n_trips[0], (n, step, k, "trips")
                kk = min(int(nt[k]), eng.trip_cap)
                assert np.array_equal(eng.trip_times[k, :kk].cpu().numpy(), o.trip_times[0, :kk]), (n, step, k, "trip times")
    n += 1
    del eng
print("agent-step fuzz ok: %d cases, %d env-decisions ended by an overflow, %.0f s" % (n, early, time.time() - t0))
