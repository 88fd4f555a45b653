"""tfx_agent_step: the Repeater (+ Remi) wrappers of the reference (traffic_test.py:27-64) fused on
the device and replayed as a HIP graph, against the same wrappers emulated tick by tick on the
oracle - including `if done: break` (an env that overflows stands still for the rest of the step)."""
import numpy as np
import pytest

from oracle.oracle import OracleEnv, live_mask

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from gym_traffic import workload as wl  # noqa: E402
from gym_traffic.core import TfxEngine  # noqa: E402


@pytest.fixture(params=["resident", "pertick", "pairs", "pairs_seg", "pairs_seg_launches"], autouse=True)
def step_path(request, monkeypatch):
    """Both forms of the fused decision: every tick in one LDS-resident launch (k_res, 3 envs per
    workgroup) and the captured sequence of per-tick kernels."""
    monkeypatch.setenv("TFX_RESIDENT", "1" if request.param == "resident" else "0")
    monkeypatch.setenv("TFX_RES_EPB", "3")
    # "pairs": two ticks per pass over the cars (k_move_tt + k_edge, k_risk inside agent steps), forced at test sizes
    monkeypatch.setenv("TFX_PAIRS", "2" if request.param.startswith("pairs") else "0")
    # (with the pairs: k_tail behind every pass, csrc/tfx_tail.hpp - "pairs_seg_launches": the separate launches, where
    # the envs k_risk sorts out take the one-tick form inside the pass -, and the env range in two halves on two streams)
    monkeypatch.setenv("TFX_TAIL", "0" if request.param == "pairs_seg_launches" else "2")
    monkeypatch.setenv("TFX_SPLIT", "2")
    # "pairs_seg*": the pass with every tile's walk split over two / four wavefronts (k_move_tts, csrc/tfx_move_tts.hpp)
    monkeypatch.setenv("TFX_TT_SEG", "2" if request.param.startswith("pairs_seg") else "0")
    monkeypatch.setenv("TFX_TT_SEGS", "4" if request.param == "pairs_seg_launches" else "2")
    yield request.param


def emulate_agent_step(orcs, tick0, action, entry, n_ticks, remi, period, archetypes=None):
    """Repeater._step + Remi._step per env on single-env oracles; returns (aobs, areward, adone).
    archetypes: the reference-shaped table of a run with heterogeneous cars (arrivals take row 0)."""
    E = len(orcs)
    r, I = orcs[0].r, orcs[0].I
    aobs = np.zeros((E, 2 * r + I), np.float32)
    arew = np.zeros((E, I), np.float32)
    adone = np.zeros(E, np.uint8)
    for k, orc in enumerate(orcs):
        total_obs = np.zeros(2 * r + I, np.float32)
        total_reward = 0
        done = False
        for t in range(n_ticks):
            tick = tick0 + t
            orc.steps[:] = tick                        # batched envs share one clock on the device
            obs, rew, d = orc.step(action[k], [wl.spawn_roads_for_tick(entry, tick, period=period)], archetypes=archetypes)
            obs, rew, done = obs[0], rew[0], bool(d[0])
            total_obs[:r] += obs[:r]
            total_obs[r:2 * r] = obs[r:2 * r]
            multiplier = 2 * obs[-2 * I:-I] - 1
            total_obs[-I:] = obs[-I:] / 100 * multiplier
            total_reward = total_reward + rew
            if done:
                break
        if remi:
            total_reward = orc.remi_reward()[0].copy()
        aobs[k], arew[k], adone[k] = total_obs, total_reward, done
    return aobs, arew, adone


@pytest.mark.parametrize("remi", [True, False])
@pytest.mark.parametrize("cap,period", [(14, 2), (34, 5)])
def test_agent_step_vs_emulated_wrappers(remi, cap, period):
    E, m, n, L, T = 7, 3, 3, 150.0, 10
    eng = TfxEngine(m, n, L, cap, n_envs=E, planes=2)
    orcs = [OracleEnv(m, n, L, cap, eng.dest, eng.phases, eng.nexts) for _ in range(E)]
    rng = np.random.RandomState(17)
    ph = rng.randint(2, size=(E, eng.I)).astype(np.int32)
    eng.reset(ph)
    for k, o in enumerate(orcs):
        o.reset(ph[k])
    eng.set_spawns(period=period)
    broke_early = 0
    for step in range(14):
        act = rng.randint(2, size=(E, eng.I)).astype(np.int32)
        eng.set_actions(act)
        tick0 = eng.tick
        aobs, arew, adone = eng.agent_step(T, remi=remi)
        eo, er, ed = emulate_agent_step(orcs, tick0, act, eng.entrypoints, T, remi, period)
        assert np.array_equal(adone.cpu().numpy(), ed), step
        assert np.array_equal(aobs.cpu().numpy(), eo), step
        assert np.array_equal(arew.cpu().numpy(), er), step
        broke_early += int(ed.sum())
        ld, lc = eng.leading.cpu().numpy(), eng.lastcar.cpu().numpy()
        x, v, _ = eng.planes_numpy()
        for k, o in enumerate(orcs):
            assert np.array_equal(ld[k], o.leading[0]) and np.array_equal(lc[k], o.lastcar[0]), (step, k)
            live = live_mask(ld[k], lc[k], cap)
            assert np.array_equal(x[k][live].view(np.int32), o.x[0][live].view(np.int32)), (step, k)
            assert np.array_equal(v[k][live].view(np.int32), o.v[0][live].view(np.int32)), (step, k)
    if cap == 14:
        assert broke_early > 0            # the `if done: break` path was exercised


def test_agent_step_graph_replay_equals_eager(monkeypatch):
    """Same run with the HIP graph disabled (TFX_GRAPH=0) gives identical outputs."""
    import os
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("TFX_GRAPH", flag)
        eng = wl.setup_engine("cfg1", envs=32)
        res = []
        for _ in range(6):
            aobs, arew, adone = eng.agent_step(10, remi=True)
            res.append((aobs.clone(), arew.clone(), adone.clone()))
        outs.append((res, eng.leading.clone(), eng.obs.clone()))
    for (a, b) in zip(outs[0][0], outs[1][0]):
        for u, v in zip(a, b):
            assert torch.equal(u, v)
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
