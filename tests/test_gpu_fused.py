"""k_res (tfx_resident.hpp): envs that fit a compute unit's LDS run all the ticks of a tfx_step /
tfx_agent_step call in one launch, the cars resident on chip.  It must be bit-identical to the
tick-by-tick kernels and to the oracle: pathological ring states (wrapped, full, empty, unsorted, cars
more than a road length past the end so that handed-off cars cascade through several roads in one
tick), per-tick action and spawn buffers, the on-device rules, rectangular grids, every capacity
class, both global car layouts, one or several envs per workgroup."""
import numpy as np
import pytest

from oracle.oracle import OracleEnv
from test_gpu_parity import (assert_engines_equal, assert_same_state, counts, load_both, oracle_like,
                             random_state)

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from gym_traffic.core import TfxEngine  # noqa: E402
from gym_traffic import workload as wl  # noqa: E402


import os  # noqa: E402


def engine_with(env, E, layout="transposed", planes=2, **cfg):
    keep = {k: os.environ.get(k) for k in env}
    os.environ.update(env)                  # read when the engine binds its buffers
    try:
        eng = TfxEngine(n_envs=E, planes=planes, layout=layout, **cfg)
    finally:
        for k, v in keep.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    return eng


def fused_engine(E, epb=2, lpr=2, **cfg):
    eng = engine_with({"TFX_RESIDENT": "1", "TFX_RES_EPB": str(epb), "TFX_RES_LPR": str(lpr)}, E, **cfg)
    assert eng.fused_ticks() == (0, True)
    return eng


def pertick_engine(E, **cfg):
    eng = engine_with({"TFX_RESIDENT": "0"}, E, **cfg)
    assert eng.fused_ticks() == (0, False)
    return eng


@pytest.mark.parametrize("m,n,C,length", [(2, 2, 10, 60.0), (3, 2, 20, 120.0), (4, 4, 34, 200.0),
                                          (2, 3, 66, 400.0), (5, 3, 12, 80.0), (1, 1, 6, 50.0),
                                          (2, 2, 130, 800.0)])
@pytest.mark.parametrize("sorted_x", [True, False])
@pytest.mark.parametrize("layout", ["transposed", "ring"])
def test_fused_random_states_vs_oracle(m, n, C, length, sorted_x, layout):
    rng = np.random.RandomState(4321 + C + int(sorted_x))
    E, T = 5, 7
    eng = fused_engine(E, epb=1 + C % 3, lpr=(1, 2, 4, 3)[(C // 2 + int(sorted_x)) % 4], layout=layout, planes=3 if layout == "ring" else 2,
                       m=m, n=n, length=length, capacity=C, rate=0.5)
    orc = oracle_like(eng)
    ran = 0
    for trial in range(5):
        x, v, w, leading, lastcar = random_state(rng, E, eng.R, C, length, crowd=rng.choice([0.3, 0.8]),
                                                 beyond=rng.choice([0.0, 0.05, 0.4, 1.6]), sorted_x=sorted_x)
        phase = rng.randint(2, size=(E, eng.I)).astype(np.int32)
        elapsed = rng.randint(0, 12, size=(E, eng.I)).astype(np.int32)
        load_both(eng, orc, x, v, w, leading, lastcar, phase, elapsed)
        eng.set_tick(60)
        orc.steps[:] = 60
        acts = rng.randint(2, size=(T, E, eng.I)).astype(np.int32)
        roads = [[rng.choice(eng.entrypoints, size=rng.randint(0, 4)).tolist() for _ in range(E)]
                 for _ in range(T)]
        eng.set_actions(acts, per_tick=True)
        eng.set_spawns(counts=np.stack([counts(eng, r) for r in roads]), per_tick=True)
        eng.step(T)
        ran += T
        done = np.zeros(E, bool)
        for t in range(T):
            done |= orc.step(acts[t], roads[t])[2].astype(bool)
        assert np.array_equal(eng.done.cpu().numpy().astype(bool), done), trial
        assert_same_state(eng, orc, "trial %d" % trial)
    assert eng.fused_ticks()[0] == ran and eng.tick == 60 + T


def test_fused_equals_tick_by_tick_on_device_rules():
    """The bench's inputs (fixed-cycle lights, periodic arrivals): 60 ticks fused in chunks of 10 ==
    60 single ticks == the same with fusing disabled; counters and done flags included."""
    E, T = 9, 60
    cfg = dict(m=4, n=4, length=200.0, capacity=34, rate=0.5)
    a = fused_engine(E, epb=2, lpr=3, **cfg)
    b = fused_engine(E, epb=1, lpr=1, **cfg)
    c = pertick_engine(E, **cfg)
    x, v, leading, lastcar = wl.prefill_one_env(4, 4, 200.0, 34, 24, 8.0)
    for eng in (a, b, c):
        eng.reset(np.zeros((E, eng.I), np.int32))
        eng.load_state(np.repeat(x[None], E, 0), np.repeat(v[None], E, 0), np.repeat(leading[None], E, 0),
                       np.repeat(lastcar[None], E, 0))
        eng.set_spawns(period=3)                 # dense arrivals: rings overflow within the run
        eng.set_actions(cycle_period=7)
        eng.reset_counters()
    for _ in range(T // 10):
        a.step(10)
    for _ in range(T):
        b.step(1)
    c.step(T)
    assert a.fused_ticks()[0] == T and b.fused_ticks()[0] == T and c.fused_ticks()[0] == 0
    assert_engines_equal(a, b)
    assert_engines_equal(c, b)
    assert a.vehicle_updates() == b.vehicle_updates() == c.vehicle_updates() > 0
    assert a.tick == b.tick == c.tick == T
    assert torch.equal(a.done_tick, b.done_tick) and int(a.done_tick.max()) > 0


def test_fused_odd_chunks_and_golden_ints(golden_cache):
    """A captured reference run fed through per-tick buffers in uneven chunks (1, 2, 3, 13, ...): the
    integers equal the reference's for the first 120 ticks, everything equals the oracle."""
    g = golden_cache("g3x3_default")
    sc = g.sc
    eng = fused_engine(1, m=sc["m"], n=sc["n"], length=sc["L"], capacity=sc["C"], rate=sc["rate"])
    orc = oracle_like(eng)
    eng.reset(g["init_phase"])
    orc.reset(g["init_phase"])
    t = 0
    for chunk in [1, 2, 3, 13, 10, 10, 7, 25, 1, 16, 32]:
        acts = g["actions"][t:t + chunk][:, None, :]
        sp = np.stack([counts(eng, [g.spawns(t + j)]) for j in range(chunk)])
        eng.set_actions(acts, per_tick=True)
        eng.set_spawns(counts=sp, per_tick=True)
        eng.step(chunk)
        for j in range(chunk):
            orc.step(g["actions"][t + j], [g.spawns(t + j)])
        t += chunk
        assert_same_state(eng, orc, "tick %d" % t)
        assert np.array_equal(eng.leading[0].cpu().numpy(), g["leading"][t])
        assert np.array_equal(eng.lastcar[0].cpu().numpy(), g["lastcar"][t])
        assert np.array_equal(eng.obs[0].cpu().numpy(), g["obs"][t])
        assert np.array_equal(eng.rewards[0].cpu().numpy(), g["rewards"][t])
    assert t == 120 and eng.fused_ticks()[0] == 120
