"""On-device Poisson arrivals + greedy controller (cfg4's closed loop): the device
stream is mirrored on the host (gym_traffic/devrng.py) and fed to the oracle together with the
greedy rule evaluated on the oracle's own cars_on_roads - bit-equal trajectories, and independent
of how the envs are sharded."""
import numpy as np
import pytest

from oracle.oracle import OracleEnv, live_mask

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from gym_traffic.core import TfxEngine  # noqa: E402
from gym_traffic.devrng import PoissonMirror, gap_table, philox4x32  # noqa: E402


@pytest.fixture(params=["resident", "pertick", "pairs"], autouse=True)
def step_path(request, monkeypatch):
    """Both producers of the on-device inputs: inside the LDS-resident kernel k_res (Poisson stream and
    greedy rule evaluated in the kernel, 2 envs per workgroup) and the per-tick kernels k_poisson /
    k_greedy."""
    monkeypatch.setenv("TFX_RESIDENT", "1" if request.param == "resident" else "0")
    monkeypatch.setenv("TFX_RES_EPB", "2")
    # "pairs": two ticks per pass over the cars (k_move_tt + k_edge, k_risk inside agent steps), forced at test sizes
    monkeypatch.setenv("TFX_PAIRS", "2" if request.param == "pairs" else "0")
    monkeypatch.setenv("TFX_TAIL", "2")        # (with the pairs: k_tail behind every pass, csrc/tfx_tail.hpp,
    monkeypatch.setenv("TFX_SPLIT", "2")       #  and the env range in two halves on two streams)
    yield request.param


def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10
    assert philox4x32(0, 0, 0, 0, 0, 0) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)
    assert philox4x32(0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF) == \
        (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)


def test_gap_table_is_the_round_exp_distribution():
    cdf = gap_table(0.8).astype(np.float64) / 2 ** 32
    rng = np.random.RandomState(0)
    gaps = np.array([round(x) for x in rng.exponential(1 / 0.8, size=200000)])
    for k in range(5):
        assert abs((gaps <= k).mean() - cdf[k]) < 4e-3


@pytest.mark.parametrize("cap,cpt,spacing", [(20, 1.4, 3), (130, 6.0, 2)])
def test_device_poisson_and_greedy_vs_oracle(cap, cpt, spacing):
    E, m, n, L, T, off = 5, 4, 3, 200.0, 90, 11
    eng = TfxEngine(m, n, L, cap, n_envs=E, planes=2, env_id_offset=off)
    orc = OracleEnv(m, n, L, cap, eng.dest, eng.phases, eng.nexts, n_envs=E)
    ph = np.zeros((E, eng.I), np.int32)
    eng.reset(ph)
    orc.reset(ph)
    eng.set_poisson(cpt, seed=0x1234ABCD5678)
    eng.set_greedy(spacing)
    mirror = PoissonMirror(cpt, 0x1234ABCD5678, eng.n_entry, range(off, off + E))
    act = np.zeros((E, eng.I), np.int32)
    total = 0
    for t in range(T):
        if t % spacing == 0:
            c = orc.cars_on_roads()                       # [E, m, n, 4]
            act = (c.reshape(E, eng.I, 4).dot([1, 1, -1, -1]) < 0).astype(np.int32)
        cnt = mirror.next_tick()
        roads = [[int(eng.entrypoints[j]) for j in range(eng.n_entry) for _ in range(cnt[k, j])] for k in range(E)]
        total += int(cnt.sum())
        eng.step(1)
        _, _, od = orc.step(act, roads)
        assert np.array_equal(eng.done.cpu().numpy(), od), t
        assert np.array_equal(eng.current_phase.cpu().numpy(), orc.current_phase), t
        assert np.array_equal(eng.leading.cpu().numpy(), orc.leading), t
        assert np.array_equal(eng.lastcar.cpu().numpy(), orc.lastcar), t
    assert total > 0.5 * cpt * T * E
    x, v, _ = eng.planes_numpy()
    ld, lc = eng.leading.cpu().numpy(), eng.lastcar.cpu().numpy()
    for k in range(E):
        live = live_mask(ld[k], lc[k], cap)
        assert np.array_equal(x[k][live].view(np.int32), orc.x[k][live].view(np.int32))
        assert np.array_equal(v[k][live].view(np.int32), orc.v[k][live].view(np.int32))
    assert np.array_equal(eng.obs.cpu().numpy(), orc.obs)
    # sharding independence: global env `off + 2` alone gives the same trajectory
    solo = TfxEngine(m, n, L, cap, n_envs=1, planes=2, env_id_offset=off + 2)
    solo.reset(ph[:1])
    solo.set_poisson(cpt, seed=0x1234ABCD5678)
    solo.set_greedy(spacing)
    solo.step(T)
    assert torch.equal(solo.leading[0], eng.leading[2]) and torch.equal(solo.obs[0], eng.obs[2])


def test_device_inputs_inside_the_agent_graph():
    """Poisson + greedy producers are part of the captured agent-step graph."""
    eng = TfxEngine(3, 3, 150.0, 34, n_envs=8, planes=2)
    ref = TfxEngine(3, 3, 150.0, 34, n_envs=8, planes=2)
    for e in (eng, ref):
        e.reset(np.zeros((8, e.I), np.int32))
        e.set_poisson(2.0, seed=7)
        e.set_greedy(3)
    for _ in range(5):
        eng.agent_step(9, remi=True)
    ref.step(45)                      # no env overflows at this load, so nothing freezes
    assert int(eng.done_tick.max()) == 0
    assert torch.equal(eng.leading, ref.leading) and torch.equal(eng.lastcar, ref.lastcar)
    assert int(eng.cars_on_roads_flat().sum()) > 100


def test_poisson_stream_continues_across_step_paths(monkeypatch):
    """Calls shorter than TFX_RES_MIN_TICKS take the per-tick kernels (k_poisson), longer ones k_res: the
    arrival stream's state (gap, car index) and the greedy controller's held action pass between them."""
    monkeypatch.setenv("TFX_RESIDENT", "1")
    monkeypatch.setenv("TFX_RES_MIN_TICKS", "4")
    E, m, n, L, cap, cpt, spacing, seed = 3, 3, 2, 120.0, 16, 1.1, 4, 0xBEEF
    eng = TfxEngine(m, n, L, cap, n_envs=E, planes=2)
    orc = OracleEnv(m, n, L, cap, eng.dest, eng.phases, eng.nexts, n_envs=E)
    ph = np.zeros((E, eng.I), np.int32)
    eng.reset(ph)
    orc.reset(ph)
    eng.set_poisson(cpt, seed=seed)
    eng.set_greedy(spacing)
    mirror = PoissonMirror(cpt, seed, eng.n_entry, range(E))
    act = np.zeros((E, eng.I), np.int32)
    t = 0
    for k in (1, 6, 2, 9, 3, 3, 11, 1, 5):
        eng.step(k)
        for _ in range(k):
            if t % spacing == 0:
                c = orc.cars_on_roads()
                act = (c.reshape(E, eng.I, 4).dot([1, 1, -1, -1]) < 0).astype(np.int32)
            cnt = mirror.next_tick()
            orc.step(act, [[int(eng.entrypoints[j]) for j in range(eng.n_entry) for _ in range(cnt[q, j])] for q in range(E)])
            t += 1
        assert np.array_equal(eng.leading.cpu().numpy(), orc.leading), (k, t)
        assert np.array_equal(eng.lastcar.cpu().numpy(), orc.lastcar), (k, t)
        assert np.array_equal(eng.obs.cpu().numpy(), orc.obs), (k, t)
    assert eng.fused_ticks()[0] == 6 + 9 + 11 + 5 and int(eng.cars_on_roads_flat().sum()) > 20


def test_vec_env_device_arrivals():
    """TrafficVecEnv(spawn='device'): the on-device Poisson stream behind the batched surface - the
    same trajectory as the engine driven directly, and sharding-independent through env_id_offset."""
    from gym_traffic.envs.vec_env import TrafficVecEnv
    venv = TrafficVecEnv(6, 3, 3, 150.0, capacity=20, spawn='device', local_cars_per_sec=0.2, seed=21)
    ph = np.zeros((6, venv.engine.I), np.int32)
    venv.reset(ph)
    ref = TfxEngine(3, 3, 150.0, 20, n_envs=2, planes=2, env_id_offset=3)
    ref.reset(ph[:2])
    ref.set_poisson(venv.cars_per_sec * venv.rate, seed=21)
    ref.set_actions(cycle_period=6)
    for _ in range(4):
        venv.agent_step(n_ticks=10, cycle_period=6)
        ref.agent_step(10)
    assert torch.equal(venv.engine.leading[3:5], ref.leading) and torch.equal(venv.engine.lastcar[3:5], ref.lastcar)
    assert int(venv.cars_on_roads().sum()) > 30


def test_poisson_generated_up_front_in_chunks(step_path):
    """tfx_step draws the arrivals of a whole call in one launch per chunk of the rows its count buffer holds (64 ticks,
    fewer for huge batches): one call of 150 ticks - three chunks, the second and third starting mid-call - leaves exactly
    the state of fifteen calls of 10 ticks, and both follow the host mirror of the stream (oracle every 50 ticks)."""
    if step_path == "resident":
        pytest.skip("the per-tick kernels' arrival stream (k_res draws inside its launch)")
    E, m, n, L, cap, cpt, spacing, seed = 5, 3, 3, 140.0, 20, 0.9, 3, 77
    a = TfxEngine(m, n, L, cap, n_envs=E, planes=2)
    b = TfxEngine(m, n, L, cap, n_envs=E, planes=2)
    orc = OracleEnv(m, n, L, cap, a.dest, a.phases, a.nexts, n_envs=E)
    ph = np.zeros((E, a.I), np.int32)
    for e in (a, b):
        e.reset(ph)
        e.set_poisson(cpt, seed=seed)
        e.set_greedy(spacing)
    orc.reset(ph)
    mirror = PoissonMirror(cpt, seed, a.n_entry, range(E))
    a.step(150)
    for _ in range(15):
        b.step(10)
    for name in ("leading", "lastcar", "obs", "rewards", "waiting", "passed_dst"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    act = np.zeros((E, a.I), np.int32)
    for t in range(150):
        if t % spacing == 0:
            c = orc.cars_on_roads()
            act = (c.reshape(E, a.I, 4).dot([1, 1, -1, -1]) < 0).astype(np.int32)
        cnt = mirror.next_tick()
        orc.step(act, [[int(a.entrypoints[j]) for j in range(a.n_entry) for _ in range(cnt[q, j])] for q in range(E)])
    assert np.array_equal(a.leading.cpu().numpy(), orc.leading) and np.array_equal(a.lastcar.cpu().numpy(), orc.lastcar)
    assert np.array_equal(a.obs.cpu().numpy(), orc.obs)
    assert int(a.cars_on_roads_flat().sum()) > 50


def test_per_tick_actions_follow_the_call_across_poisson_chunks(step_path):
    """A per-tick action buffer is indexed by the tick of the WHOLE call while the arrivals are drawn in chunks of the
    rows the count buffer holds: 150 ticks in one call (three chunks) == fifteen calls of 10 ticks fed rows 10k..10k+9,
    == the oracle under the host mirror of the stream (round-3 advisor finding: chunks restarted the action row at 0)."""
    if step_path == "resident":
        pytest.skip("the per-tick kernels' arrival stream (k_res draws inside its launch)")
    E, m, n, L, cap, cpt, seed, T = 4, 3, 3, 140.0, 20, 0.9, 91, 150
    rng = np.random.RandomState(5)
    acts = np.repeat(rng.randint(2, size=(T // 5, E, m * n)), 5, axis=0).astype(np.int32)      # a new action every 5 ticks
    a = TfxEngine(m, n, L, cap, n_envs=E, planes=2)
    b = TfxEngine(m, n, L, cap, n_envs=E, planes=2)
    orc = OracleEnv(m, n, L, cap, a.dest, a.phases, a.nexts, n_envs=E)
    ph = np.zeros((E, a.I), np.int32)
    for e in (a, b):
        e.reset(ph)
        e.set_poisson(cpt, seed=seed)
    orc.reset(ph)
    a.set_actions(torch.as_tensor(acts).to(a.device), per_tick=True)
    a.step(T)
    for k in range(T // 10):
        b.set_actions(torch.as_tensor(acts[10 * k:10 * k + 10]).to(b.device), per_tick=True)
        b.step(10)
    for name in ("leading", "lastcar", "obs", "rewards", "waiting", "passed_dst"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    mirror = PoissonMirror(cpt, seed, a.n_entry, range(E))
    for t in range(T):
        cnt = mirror.next_tick()
        orc.step(acts[t], [[int(a.entrypoints[j]) for j in range(a.n_entry) for _ in range(cnt[q, j])] for q in range(E)])
    assert np.array_equal(a.leading.cpu().numpy(), orc.leading) and np.array_equal(a.lastcar.cpu().numpy(), orc.lastcar)
    assert np.array_equal(a.obs.cpu().numpy(), orc.obs)


def golden_regular_counts(g):
    """cars per tick of a reference-captured run under the `regular` generator, and its cars_per_tick"""
    off = g["spawn_off"]
    return np.diff(off).astype(np.int64), float(g["cars_per_sec"]) * g.sc["rate"]


@pytest.mark.parametrize("name", ["g2x2_s0_reg_c20", "g2x2_s2_reg_c10"])
def test_device_regular_generator_counts_and_roads(name, golden_cache, step_path):
    """tfx_set_regular (the reference's `regular`, traffic_env.py:167-176, on the device): the cars made per tick are
    EXACTLY the reference's captured schedule (ceil(cpt) every round(1 / cpt) ticks), each car's entry road is the host
    mirror's draw bit for bit (gym_traffic/devrng.py RegularMirror), and the trajectory equals the oracle's under the
    mirrored arrivals - tick by tick, in multi-tick calls and through fused decisions."""
    from gym_traffic.devrng import RegularMirror
    g = golden_cache(name)
    sc = g.sc
    per_tick, cpt = golden_regular_counts(g)
    E, off, seed, T = 3, 5, 0xABCDEF12345, 60
    eng = TfxEngine(sc["m"], sc["n"], sc["L"], sc["C"], n_envs=E, planes=2, rate=sc["rate"], env_id_offset=off)
    orc = OracleEnv(sc["m"], sc["n"], sc["L"], sc["C"], eng.dest, eng.phases, eng.nexts, n_envs=E)
    ph = np.zeros((E, eng.I), np.int32)
    eng.reset(ph)
    orc.reset(ph)
    eng.set_regular(cpt, seed=seed)
    eng.set_actions(cycle_period=7)
    mirror = RegularMirror(cpt, seed, eng.n_entry, range(off, off + E))
    t = 0
    for k in (1, 4, 1, 7, 2, 10, 3, 1, 9, 6, 16):
        eng.step(k)
        for _ in range(k):
            cnt = mirror.next_tick()
            assert (cnt.sum(axis=1) == per_tick[t]).all(), t          # the reference's own count for this tick
            act = (((t + (np.arange(off, off + E) % 7)) // 7) & 1).astype(np.int32)
            orc.step(np.repeat(act[:, None], eng.I, axis=1),
                     [[int(eng.entrypoints[j]) for j in range(eng.n_entry) for _ in range(cnt[q, j])] for q in range(E)])
            t += 1
        assert np.array_equal(eng.leading.cpu().numpy(), orc.leading), (k, t)
        assert np.array_equal(eng.lastcar.cpu().numpy(), orc.lastcar), (k, t)
        assert np.array_equal(eng.obs.cpu().numpy(), orc.obs), (k, t)
    assert t == T and int(eng.cars_on_roads_flat().sum()) > 10
    # sharding independence: global env `off + 1` alone
    solo = TfxEngine(sc["m"], sc["n"], sc["L"], sc["C"], n_envs=1, planes=2, rate=sc["rate"], env_id_offset=off + 1)
    solo.reset(ph[:1])
    solo.set_regular(cpt, seed=seed)
    solo.set_actions(cycle_period=7)
    solo.step(T)
    assert torch.equal(solo.leading[0], eng.leading[1]) and torch.equal(solo.obs[0], eng.obs[1])


def test_vec_env_regular_device_arrivals():
    """TrafficVecEnv(spawn='regular_device'): the batched surface over tfx_set_regular, fused decisions included."""
    from gym_traffic.envs.vec_env import TrafficVecEnv
    venv = TrafficVecEnv(6, 3, 3, 150.0, capacity=20, spawn='regular_device', local_cars_per_sec=0.2, seed=4)
    ph = np.zeros((6, venv.engine.I), np.int32)
    venv.reset(ph)
    ref = TfxEngine(3, 3, 150.0, 20, n_envs=2, planes=2, env_id_offset=3)
    ref.reset(ph[:2])
    ref.set_regular(venv.cars_per_sec * venv.rate, seed=4)
    ref.set_actions(cycle_period=6)
    for _ in range(4):
        venv.agent_step(n_ticks=10, cycle_period=6)
        ref.step(10)
    assert int(venv.engine.done_tick.max()) == 0           # (nothing overflowed, so nothing froze)
    assert torch.equal(venv.engine.leading[3:5], ref.leading) and torch.equal(venv.engine.lastcar[3:5], ref.lastcar)
    import math
    cpt = venv.cars_per_sec * venv.rate
    made = 40 // round(1 / cpt) * math.ceil(cpt) if round(1 / cpt) else 40 * math.ceil(cpt)
    assert int(venv.cars_on_roads().sum()) > 0 and made > 0
