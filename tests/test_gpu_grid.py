"""k_grid (csrc/tfx_grid.hpp): every tick of a tfx_step call in one cooperative launch, a workgroup per tile with the
tile's cars resident in LDS and grid barriers between the phases of a tick.  The handle takes it on its own when its
tiles cannot fill the chip with two-tick passes and all fit at once (BASELINE config 5: one 64x64 grid); here it is also
forced at test sizes (TFX_GRID=2, k_res off) and must be bit-identical to the tick-by-tick kernels and to the oracle:
pathological ring states (wrapped, full, empty, unsorted, cars several road lengths past the end - the literal serial
handoff inside the launch -, more than two pops per road and tick), per-tick action and spawn buffers, the on-device
rules, Poisson arrivals + the greedy controller, one and several tiles per workgroup, call lengths from 2 ticks on."""
import numpy as np
import pytest

from test_gpu_parity import (assert_engines_equal, assert_same_state, counts, load_both, oracle_like,
                             random_state)
from test_gpu_fused import engine_with

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from gym_traffic import workload as wl  # noqa: E402
from gym_traffic.core import TfxEngine  # noqa: E402
from gym_traffic.devrng import PoissonMirror  # noqa: E402


def grid_engine(E, **cfg):
    return engine_with({"TFX_RESIDENT": "0", "TFX_GRID": "2", "TFX_PAIRS": "0"}, E, **cfg)


def pertick_engine(E, **cfg):
    return engine_with({"TFX_RESIDENT": "0", "TFX_GRID": "0", "TFX_PAIRS": "0"}, E, **cfg)


@pytest.mark.parametrize("m,n,C,length", [(2, 2, 10, 60.0), (3, 2, 20, 120.0), (4, 4, 34, 200.0), (2, 3, 66, 400.0),
                                          (5, 3, 12, 80.0), (1, 1, 6, 50.0), (2, 2, 130, 800.0), (6, 6, 34, 150.0)])
@pytest.mark.parametrize("sorted_x", [True, False])
def test_grid_random_states_vs_oracle(m, n, C, length, sorted_x):
    rng = np.random.RandomState(9753 + C + int(sorted_x))
    E = 3
    eng = grid_engine(E, m=m, n=n, length=length, capacity=C, rate=0.5)
    orc = oracle_like(eng)
    for trial, T in enumerate([3, 4, 7, 2, 5, 11, 6]):
        x, v, w, leading, lastcar = random_state(rng, E, eng.R, C, length, crowd=rng.choice([0.3, 0.8]),
                                                 beyond=rng.choice([0.0, 0.05, 0.4, 1.6]), sorted_x=sorted_x)
        if trial in (4, 5):
            # leave the fast-division domain: enormous and denormal speeds, exact-zero gap denominators
            v[rng.rand(*v.shape) < 0.02] = 3e7
            v[rng.rand(*v.shape) < 0.02] = 1e-30
            pick = rng.rand(*x[:, :, 2:].shape) < 0.05
            x[:, :, 2:][pick] = (x[:, :, 1:-1] - np.float32(4.0))[pick]
            np.put_along_axis(x, leading[:, :, None].astype(np.int64), np.inf, axis=2)
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
        assert eng.step_kernel() == "k_grid", trial
        done = np.zeros(E, bool)
        for t in range(T):
            done |= orc.step(acts[t], roads[t])[2].astype(bool)
        assert np.array_equal(eng.done.cpu().numpy().astype(bool), done), trial
        assert_same_state(eng, orc, "trial %d (%d ticks)" % (trial, T))


@pytest.mark.parametrize("chunk", [2, 3, 10, 25])
def test_grid_equals_tick_by_tick_on_device_rules(chunk):
    """The bench's inputs (fixed-cycle lights, periodic arrivals), dense enough that rings overflow: 60 ticks in calls
    of `chunk` ticks == the same on the tick-by-tick kernels; counters, rewards and done flags included."""
    E, T = 4, 60
    cfg = dict(m=4, n=4, length=200.0, capacity=34, rate=0.5)
    a = grid_engine(E, **cfg)
    c = pertick_engine(E, **cfg)
    x, v, leading, lastcar = wl.prefill_one_env(4, 4, 200.0, 34, 24, 8.0)
    for eng in (a, c):
        eng.reset(np.zeros((E, eng.I), np.int32))
        eng.load_state(np.repeat(x[None], E, 0), np.repeat(v[None], E, 0), np.repeat(leading[None], E, 0),
                       np.repeat(lastcar[None], E, 0))
        eng.set_spawns(period=3)
        eng.set_actions(cycle_period=7)
        eng.reset_counters()
    done = 0
    while done < T:
        k = min(chunk, T - done)
        a.step(k)
        c.step(k)
        done += k
        assert_engines_equal(a, c)                              # (rewards and done flags included)
    assert a.step_kernel() == "k_grid" and c.step_kernel() != "k_grid"
    assert a.vehicle_updates() == c.vehicle_updates()
    assert int(a.done_tick.max()) > 0                          # rings did overflow on the way


@pytest.mark.parametrize("E,m,n,C", [(1, 6, 6, 66), (2, 4, 4, 130), (8, 2, 2, 20)])
def test_grid_poisson_and_greedy_closed_loop_vs_oracle(E, m, n, C):
    """The closed loop of BASELINE config 5 at test size: arrivals drawn on the device (Philox, mirrored on the host), the
    greedy controller deciding every third tick from cars_on_roads AFTER the handoff (its own pair of barriers)."""
    length = 200.0
    eng = grid_engine(E, m=m, n=n, length=length, capacity=C, rate=0.5)
    orc = oracle_like(eng)
    ph = np.zeros((E, eng.I), np.int32)
    eng.reset(ph)
    orc.reset(ph)
    cpt, seed, spacing = 1.7, 4321, 3
    eng.set_poisson(cpt, seed=seed)
    eng.set_greedy(spacing)
    mirror = PoissonMirror(cpt, seed, eng.n_entry, range(E))
    act = np.zeros((E, eng.I), np.int32)
    tick = 0
    for T in (5, 12, 2, 30, 7):
        eng.step(T)
        assert eng.step_kernel() == "k_grid"
        for _ in range(T):
            if tick % spacing == 0:                            # greedy.py:14-16, decided from the state before the tick
                act = (orc.cars_on_roads().reshape(E, eng.I, 4).dot([1, 1, -1, -1]) < 0).astype(np.int32)
            cnt = mirror.next_tick()
            roads = [[int(eng.entrypoints[j]) for j in range(eng.n_entry) for _ in range(cnt[k, j])] for k in range(E)]
            orc.step(act, roads)
            tick += 1
        assert_same_state(eng, orc, "after %d ticks" % tick)
    assert int(orc.cars_on_roads().sum()) > 0


def test_grid_several_tiles_per_workgroup():
    """More tiles than workgroups can be resident at once: a workgroup owns two or three tiles (at cfg4 four of the 256
    do).  128-car rings: 66.5 KB of LDS per tile, so every tile past the 512th... is not reachable at test size; instead
    the handle is told the chip has 2 compute units (TFX_GRID_CUS) - the ownership loop is the same code."""
    E, T = 2, 9
    cfg = dict(m=5, n=5, length=200.0, capacity=130, rate=0.5)
    a = engine_with({"TFX_RESIDENT": "0", "TFX_GRID": "2", "TFX_PAIRS": "0", "TFX_GRID_CUS": "2"}, E, **cfg)
    c = pertick_engine(E, **cfg)
    rng = np.random.RandomState(31)
    x, v, w, leading, lastcar = random_state(rng, E, a.R, 130, 200.0, crowd=0.6, beyond=0.4, sorted_x=True)
    phase = rng.randint(2, size=(E, a.I)).astype(np.int32)
    elapsed = rng.randint(0, 12, size=(E, a.I)).astype(np.int32)
    for eng in (a, c):
        eng.reset(phase)
        eng.load_state(x, v, leading, lastcar)
        eng.set_spawns(period=2)
        eng.set_actions(cycle_period=5)
    for _ in range(3):
        a.step(T)
        c.step(T)
        assert_engines_equal(a, c)
    assert a.step_kernel() == "k_grid"


def test_grid_is_the_default_for_one_big_env():
    """The handle's own choice: one 16x16 env with 66-slot rings (34 tiles: no two-tick passes, too big for k_res) runs
    its tfx_step calls on k_grid; a batch that fills the chip does not."""
    eng = TfxEngine(16, 16, 400.0, 66, n_envs=1, planes=2)
    eng.reset(np.zeros((1, eng.I), np.int32))
    eng.set_spawns(period=4)
    eng.set_actions(cycle_period=10)
    eng.step(10)
    assert eng.step_kernel() == "k_grid"
    eng.step(1)                                               # a single tick stays on the per-tick kernels
    assert eng.step_kernel() != "k_grid"
    ref = pertick_engine(1, m=16, n=16, length=400.0, capacity=66)
    ref.reset(np.zeros((1, ref.I), np.int32))
    ref.set_spawns(period=4)
    ref.set_actions(cycle_period=10)
    ref.step(11)
    assert_engines_equal(eng, ref)


@pytest.mark.parametrize("lcps,prefill", [(0.12, 0), (0.12, 96)])
def test_cfg4_real_size_closed_loop_on_k_grid_vs_oracle(lcps, prefill):
    """BASELINE config 5 at its real size through the handle's own choice for it: GridRoad(64, 64, 800), CAPACITY = 130,
    one env - 260 tiles on 256 workgroups -, on-device Poisson arrivals and the greedy controller every 3 ticks, in
    tfx_step calls of 10 ticks (the oracle fed by the host mirror of the device stream), from an empty start and from the
    benchmark's prefill (1.6 M cars: handoffs across all 4096 intersections)."""
    from oracle.oracle import OracleEnv
    m = n = 64
    C, L, spacing, seed = 130, 800.0, 3, 1234
    cpt = lcps * m * 4 * 0.5
    eng = TfxEngine(m, n, L, C, n_envs=1, planes=2)
    orc = OracleEnv(m, n, L, C, eng.dest, eng.phases, eng.nexts, n_envs=1)
    ph = np.zeros((1, eng.I), np.int32)
    eng.reset(ph)
    orc.reset(ph)
    if prefill:
        x0, v0, ld0, lc0 = wl.prefill_one_env(m, n, L, C, prefill, 8.0)
        eng.load_state(x0[None], v0[None], ld0[None], lc0[None])
        orc.load_planes(0, x0, v0, np.zeros_like(x0), ld0, lc0)
    eng.set_poisson(cpt, seed=seed)
    eng.set_greedy(spacing)
    mirror = PoissonMirror(cpt, seed, eng.n_entry, [0])
    entry = np.asarray(eng.entrypoints)
    act = np.zeros((1, eng.I), np.int32)
    t = 0
    for T in (10, 10, 7, 10, 3, 10, 10):
        eng.step(T)
        assert eng.step_kernel() == "k_grid"
        done = np.zeros(1, bool)
        for _ in range(T):
            if t % spacing == 0:
                act = (orc.cars_on_roads().reshape(1, eng.I, 4).dot([1, 1, -1, -1]) < 0).astype(np.int32)
            cnt = mirror.next_tick()
            done |= orc.step(act, [np.repeat(entry, cnt[0])], nthreads=8)[2].astype(bool)
            t += 1
        assert np.array_equal(eng.done.cpu().numpy().astype(bool), done), t
        assert np.array_equal(eng.leading.cpu().numpy(), orc.leading), t
        assert np.array_equal(eng.lastcar.cpu().numpy(), orc.lastcar), t
        assert np.array_equal(eng.obs.cpu().numpy(), orc.obs), t
        assert np.array_equal(eng.rewards.cpu().numpy(), orc.rewards), t
    assert_same_state(eng, orc, "cfg4 on k_grid, %d ticks" % t)
