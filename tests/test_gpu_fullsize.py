"""Headline-size (cfg2: 4096 envs x 16x16 x 64-car roads) properties that need no per-car
reference: conservation of cars, sortedness, determinism (a race would show as drift between two
runs), independence of an env from the batch it is in, agreement with the oracle on a sample of
envs, and the same for the cfg4-shaped 2-wave-per-road kernel."""
import numpy as np
import pytest

from oracle.oracle import OracleEnv, live_mask

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from gym_traffic import workload as wl  # noqa: E402


def oracle_for(eng, c, envs, first_env, validate=False):
    orc = OracleEnv(c["m"], c["n"], c["length"], c["capacity"], eng.dest, eng.phases, eng.nexts, n_envs=envs, validate=validate)
    orc.reset(np.zeros(orc.I, np.int32))
    x, v, leading, lastcar = wl.prefill_one_env(c["m"], c["n"], c["length"], c["capacity"], c["prefill"], c["gap"])
    for k in range(envs):
        orc.load_planes(k, x, v, np.zeros_like(x), leading, lastcar)
    return orc, np.arange(first_env, first_env + envs)


def step_oracle(orc, ids, eng, t, threads=8):
    roads = wl.spawn_roads_for_tick(eng.entrypoints, t)
    return orc.step(wl.cycle_actions(ids, orc.I, t), [roads] * len(ids), nthreads=threads)


@pytest.mark.parametrize("layout", ["transposed", "ring"])
def test_cfg2_full_size_properties(layout):
    c = wl.CONFIGS["cfg2"]
    T = 45
    eng = wl.setup_engine("cfg2", layout=layout)
    E, R, C = eng.E, eng.R, eng.C
    assert (E, R, C) == (4096, 1088, 66)
    sample = [0, 1, 19, 20, 2047, 4095]                  # env ids checked against the oracle
    orcs = [oracle_for(eng, c, 1, k) for k in sample]
    small = wl.setup_engine("cfg2", envs=3, env_id_offset=19, layout=layout)   # global envs 19, 20, 21 in a tiny batch
    cars = int(eng.cars_on_roads_flat().sum())
    assert cars == E * R * 48
    eng.reset_counters()
    upd = 0
    for t in range(T):
        before = eng.cars_on_roads_flat().to(torch.int64).sum().item()
        eng.step(1)
        small.step(1)
        for (orc, ids) in orcs:
            step_oracle(orc, ids, eng, t, threads=1)
        # conservation: cars after = cars before + accepted spawns - cars that left the map
        after = eng.cars_on_roads_flat().to(torch.int64).sum().item()
        spawned = len(wl.spawn_roads_for_tick(eng.entrypoints, t)) * E
        assert after <= before + spawned
        upd += before          # lower bound of the device counter (spawned cars add to it)
    assert eng.vehicle_updates() >= upd
    ld, lc = eng.leading.cpu().numpy(), eng.lastcar.cpu().numpy()
    assert ld.min() >= 1 and lc.min() >= 1 and ld.max() <= C - 1 and lc.max() <= C - 1
    # the six sampled envs equal the oracle bit for bit (ints + live x, v)
    for (orc, ids), k in zip(orcs, sample):
        assert np.array_equal(ld[k], orc.leading[0]) and np.array_equal(lc[k], orc.lastcar[0]), k
        assert np.array_equal(eng.obs[k].cpu().numpy(), orc.obs[0]), k
        assert np.array_equal(eng.rewards[k].cpu().numpy(), orc.rewards[0]), k
        live = live_mask(ld[k], lc[k], C)
        xk, vk = eng.x[k].cpu().numpy(), eng.v[k].cpu().numpy()
        assert np.array_equal(xk[live].view(np.int32), orc.x[0][live].view(np.int32)), k
        assert np.array_equal(vk[live].view(np.int32), orc.v[0][live].view(np.int32)), k
    # an env's trajectory does not depend on the batch it is stepped in
    assert torch.equal(small.leading, eng.leading[19:22]) and torch.equal(small.obs, eng.obs[19:22])
    # sortedness: x never increases from head to tail (checked on a slice of envs)
    xs = eng.x[:64].cpu().numpy()
    for k in range(0, 64, 9):
        for e in range(0, R, 37):
            s, order = int(ld[k, e]), []
            while s != int(lc[k, e]):
                s = s + 1 if s + 1 < C else 1
                order.append(xs[k, e, s])
            assert all(a >= b for a, b in zip(order, order[1:])), (k, e)


def test_cfg2_full_size_two_tick_passes_vs_oracle():
    """The headline launch as the benchmark runs it: multi-tick calls, i.e. two-tick passes (k_move_tt + k_edge) over
    all 4096 envs.  Sampled envs equal the oracle bit for bit after calls of even and odd lengths; an env stepped
    in a tiny batch (51 tiles: the pairs with every tile's walk split over eight wavefronts, k_move_tts, its odd ticks on
    k_move_ts) has the same trajectory."""
    c = wl.CONFIGS["cfg2"]
    eng = wl.setup_engine("cfg2")
    C = eng.C
    sample = [0, 7, 1023, 2048, 4095]
    orcs = [oracle_for(eng, c, 1, k) for k in sample]
    small = wl.setup_engine("cfg2", envs=3, env_id_offset=2047)
    t = 0
    for chunk in [2, 10, 5, 8, 1, 6]:
        eng.step(chunk)
        small.step(chunk)
        for _ in range(chunk):
            for (orc, ids) in orcs:
                step_oracle(orc, ids, eng, t, threads=1)
            t += 1
        ld, lc = eng.leading.cpu().numpy(), eng.lastcar.cpu().numpy()
        for (orc, ids), k in zip(orcs, sample):
            assert np.array_equal(ld[k], orc.leading[0]) and np.array_equal(lc[k], orc.lastcar[0]), (t, k)
            assert np.array_equal(eng.obs[k].cpu().numpy(), orc.obs[0]), (t, k)
            assert np.array_equal(eng.waiting[k].cpu().numpy(), orc.waiting[0]), (t, k)
    assert eng.pair_ticks() == 2 + 10 + 4 + 8 + 6 and small.pair_ticks() == eng.pair_ticks()
    assert small.step_kernel() == "k_move_tts" and eng.step_kernel() == "k_move_tt"
    # ... behind every pass one k_tail launch, the batch in two halves on two streams (the defaults at this size)
    assert eng.tail_ticks() == eng.pair_ticks() and eng.split_ticks() == 2 + 10 + 5 + 8 + 6
    for (orc, ids), k in zip(orcs, sample):
        live = live_mask(ld[k], lc[k], C)
        xk, vk = eng.x[k].cpu().numpy(), eng.v[k].cpu().numpy()
        assert np.array_equal(xk[live].view(np.int32), orc.x[0][live].view(np.int32)), k
        assert np.array_equal(vk[live].view(np.int32), orc.v[0][live].view(np.int32)), k
    assert torch.equal(small.leading, eng.leading[2047:2050]) and torch.equal(small.obs, eng.obs[2047:2050])
    lds, lcs = small.leading.cpu().numpy(), small.lastcar.cpu().numpy()
    for j in range(3):
        live = live_mask(lds[j], lcs[j], C)
        assert np.array_equal(small.x[j].cpu().numpy()[live].view(np.int32), eng.x[2047 + j].cpu().numpy()[live].view(np.int32))


def test_cfg2_full_size_validate_mode_in_pairs_vs_oracle():
    """Validate mode at the headline size: the spawn-tick plane through the W forms of the pass and of k_tail, both halves
    of the split; sampled envs equal the oracle in every integer, every live car's x / v / w and the trip log."""
    c = wl.CONFIGS["cfg2"]
    eng = wl.setup_engine("cfg2", validate=True)
    C = eng.C
    sample = [0, 2047, 2048, 4095]
    orcs = [oracle_for(eng, c, 1, k, validate=True) for k in sample]
    t = 0
    for chunk in [2, 9, 12, 1, 16]:
        eng.step(chunk)
        for _ in range(chunk):
            for (orc, ids) in orcs:
                step_oracle(orc, ids, eng, t, threads=1)
            t += 1
    assert eng.pair_ticks() == 2 + 8 + 12 + 16 and eng.tail_ticks() == eng.pair_ticks() and eng.split_ticks() == 2 + 9 + 12 + 16
    ld, lc = eng.leading.cpu().numpy(), eng.lastcar.cpu().numpy()
    nt = eng.n_trips.cpu().numpy()
    assert int(nt.sum()) > 4096 * 10
    for (orc, ids), k in zip(orcs, sample):
        assert np.array_equal(ld[k], orc.leading[0]) and np.array_equal(lc[k], orc.lastcar[0]), k
        assert np.array_equal(eng.obs[k].cpu().numpy(), orc.obs[0]) and np.array_equal(eng.waiting[k].cpu().numpy(), orc.waiting[0]), k
        live = live_mask(ld[k], lc[k], C)
        for got, want in ((eng.x[k], orc.x[0]), (eng.v[k], orc.v[0]), (eng.w[k], orc.w[0])):
            assert np.array_equal(got.cpu().numpy()[live].view(np.int32), want[live].view(np.int32)), k
        n = int(nt[k])
        assert n == int(orc.n_trips[0]) and 0 < n <= eng.trip_cap
        assert np.array_equal(eng.trip_times[k, :n].cpu().numpy(), orc.trip_times[0, :n]), k


@pytest.mark.parametrize("layout", ["ring", "transposed"])
def test_cfg2_determinism_two_runs(layout):
    a = wl.setup_engine("cfg2", envs=512, layout=layout)
    b = wl.setup_engine("cfg2", envs=512, layout=layout)
    a.step(60)
    for _ in range(6):
        b.step(10)
    for name in ("leading", "lastcar", "obs", "rewards", "waiting", "passed_dst"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    ld, lc = a.leading.cpu().numpy(), a.lastcar.cpu().numpy()
    xa, xb = a.xv.cpu().numpy(), b.xv.cpu().numpy()
    for k in range(0, 512, 37):
        live = live_mask(ld[k], lc[k], a.C)
        assert np.array_equal(xa[k][live].view(np.int32), xb[k][live].view(np.int32))


@pytest.mark.parametrize("layout", ["ring", "transposed"])
def test_cfg4_shape_two_waves_per_road_vs_oracle(layout):
    """64-wide rings do not fit one wavefront at 128 cars/road: cfg4's kernel (k_move<2>) on a
    smaller grid with the same CAPACITY = 130, against the oracle."""
    c = dict(wl.CONFIGS["cfg4"], m=4, n=4, envs=3)
    wl.CONFIGS["_cfg4_small"] = c
    try:
        eng = wl.setup_engine("_cfg4_small", layout=layout)
        orc, ids = oracle_for(eng, c, 3, 0)
        for t in range(40):
            eng.step(1)
            step_oracle(orc, ids, eng, t, threads=3)
        ld, lc = eng.leading.cpu().numpy(), eng.lastcar.cpu().numpy()
        assert np.array_equal(ld, orc.leading) and np.array_equal(lc, orc.lastcar)
        assert np.array_equal(eng.obs.cpu().numpy(), orc.obs)
        x, v, _ = eng.planes_numpy()
        for k in range(3):
            live = live_mask(ld[k], lc[k], eng.C)
            assert np.array_equal(x[k][live].view(np.int32), orc.x[k][live].view(np.int32))
            assert np.array_equal(v[k][live].view(np.int32), orc.v[k][live].view(np.int32))
    finally:
        del wl.CONFIGS["_cfg4_small"]


@pytest.mark.parametrize("layout", ["ring", "transposed"])
def test_cfg4_capacity_two_pass_tiled_kernel_vs_oracle(layout):
    """CAPACITY = 130 (cfg4's 128-car roads) at a batch large enough for the tiled LDS-DMA kernel:
    a wavefront takes each road in two passes of 64 cars.  16x16 grid x 16 envs, heavy prefill so
    both passes are busy, against the oracle bit for bit."""
    c = dict(wl.CONFIGS["cfg4"], m=16, n=16, envs=16, prefill=100)
    wl.CONFIGS["_cfg4_tiled"] = c
    try:
        eng = wl.setup_engine("_cfg4_tiled", layout=layout)
        assert eng.E * eng.R >= 64 * 256
        orc, ids = oracle_for(eng, c, 16, 0)
        for t in range(36):
            eng.step(1)
            _, _, od = step_oracle(orc, ids, eng, t, threads=8)
            assert np.array_equal(eng.done.cpu().numpy(), od), t
        ld, lc = eng.leading.cpu().numpy(), eng.lastcar.cpu().numpy()
        assert np.array_equal(ld, orc.leading) and np.array_equal(lc, orc.lastcar)
        assert np.array_equal(eng.obs.cpu().numpy(), orc.obs)
        assert np.array_equal(eng.waiting.cpu().numpy(), orc.waiting)
        x, v, _ = eng.planes_numpy()
        for k in range(16):
            live = live_mask(ld[k], lc[k], eng.C)
            assert np.array_equal(x[k][live].view(np.int32), orc.x[k][live].view(np.int32)), k
            assert np.array_equal(v[k][live].view(np.int32), orc.v[k][live].view(np.int32)), k
        assert int((lc != ld).sum()) > 1000 and int(eng.cars_on_roads_flat().max()) > 64
    finally:
        del wl.CONFIGS["_cfg4_tiled"]


@pytest.mark.parametrize("layout", ["transposed", "ring"])
@pytest.mark.parametrize("lcps,prefill,T", [(0.12, 0, 150), (0.02, 0, 150), (0.12, 96, 60)])
def test_cfg4_real_size_closed_loop_vs_oracle(layout, lcps, prefill, T):
    """BASELINE config 5 at its real size: GridRoad(64, 64, 800), CAPACITY = 130, one env, empty start,
    on-device Poisson arrivals (local_cars_per_sec 0.12 -> 15.36 cars/tick nominal, which the
    reference's whole-tick gap rounding turns into bursts of thousands: traffic_env.py:160-164) and the
    greedy controller every 3 ticks (greedy.py:14-16 on cars_on_roads, traffic_env.py:255-257), 150
    ticks.  The oracle is fed by the host mirror of the device stream and the greedy rule evaluated
    on its own counts; ring indices, obs, rewards, done every tick, every live car at the end.
    From an empty start the cars stay on the 256 entry roads for the first ~115 ticks, so a third case
    starts from the benchmark's prefill (96 cars on each of the 16 640 roads, 1.6 M cars): handoffs
    across all 4096 intersections and the greedy rule deciding on real queues."""
    from gym_traffic.core import TfxEngine
    from gym_traffic.devrng import PoissonMirror
    m = n = 64
    C, L, spacing, seed = 130, 800.0, 3, 1234
    cpt = lcps * m * 4 * 0.5
    eng = TfxEngine(m, n, L, C, n_envs=1, planes=2, layout=layout)
    assert (eng.R, eng.I) == (16640, 4096)
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
    arrived = overflow_ticks = passed = switched = 0
    for t in range(T):
        if t % spacing == 0:
            c = orc.cars_on_roads()
            new = (c.reshape(1, eng.I, 4).dot([1, 1, -1, -1]) < 0).astype(np.int32)
            switched += int((new != act).sum())
            act = new
        cnt = mirror.next_tick()
        arrived += int(cnt.sum())
        eng.step(1)
        _, _, od = orc.step(act, [np.repeat(entry, cnt[0])], nthreads=8)
        overflow_ticks += int(od[0])
        passed += int(orc.passed.sum())
        assert np.array_equal(eng.done.cpu().numpy(), od), t
        assert np.array_equal(eng.leading.cpu().numpy(), orc.leading), t
        assert np.array_equal(eng.lastcar.cpu().numpy(), orc.lastcar), t
        assert np.array_equal(eng.obs.cpu().numpy(), orc.obs), t
        assert np.array_equal(eng.rewards.cpu().numpy(), orc.rewards), t
    assert arrived > 0.5 * cpt * T
    assert (overflow_ticks > 40) == (lcps == 0.12)        # the nominal rate saturates the entry roads
    if prefill:
        assert passed > 1000 and switched > 100, (passed, switched)
    ld, lc = eng.leading.cpu().numpy(), eng.lastcar.cpu().numpy()
    x, v, _ = eng.planes_numpy()
    live = live_mask(ld[0], lc[0], C)
    assert int(live.sum()) > (300 if not prefill else 1000000)
    assert np.array_equal(x[0][live].view(np.int32), orc.x[0][live].view(np.int32))
    assert np.array_equal(v[0][live].view(np.int32), orc.v[0][live].view(np.int32))
    assert np.array_equal(eng.waiting.cpu().numpy(), orc.waiting)


def test_cfg4_sixteen_envs_properties():
    """cfg4 x 16 envs, closed loop on the device: an env's trajectory does not depend on the batch it
    runs in (global env 5 and 15 alone give the same rings), ring indices stay in range, cars are
    conserved up to arrivals."""
    from gym_traffic.core import TfxEngine
    m = n = 64
    cpt = 0.12 * m * 4 * 0.5

    def make(E, off):
        e = TfxEngine(m, n, 800.0, 130, n_envs=E, planes=2, env_id_offset=off)
        e.reset(np.zeros((1, e.I), np.int32))
        e.set_poisson(cpt, seed=1234)
        e.set_greedy(3)
        return e
    eng = make(16, 0)
    solos = {k: make(1, k) for k in (5, 15)}
    for _ in range(6):
        eng.step(10)
        for s in solos.values():
            s.step(10)
    ld, lc = eng.leading.cpu().numpy(), eng.lastcar.cpu().numpy()
    assert ld.min() >= 1 and lc.min() >= 1 and ld.max() <= 129 and lc.max() <= 129
    for k, s in solos.items():
        assert torch.equal(s.leading[0], eng.leading[k]) and torch.equal(s.lastcar[0], eng.lastcar[k]), k
        assert torch.equal(s.obs[0], eng.obs[k]), k
    occ = eng.cars_on_roads_flat().cpu().numpy()
    assert occ.max() == 128 and (occ.sum(1) > 10000).all()
    # (after 60 ticks every entry road is full and no car has reached a second road yet, so the counts
    # agree; the envs still differ car by car - each has its own arrival stream)
    xv = eng.xv
    assert not torch.equal(xv[0], xv[1])


def test_cfg1_full_batch_default_resident_packing_vs_oracle(monkeypatch):
    """BASELINE config 2 as the benchmark runs it: 1024 envs of 4x4 x 32-car roads through k_res with the DEFAULT
    env-per-workgroup packing (no TFX_RES_EPB / TFX_RES_LPR / TFX_RESIDENT override), 60 ticks in uneven
    tfx_step(n) calls and then two fused 10-tick agent decisions (Repeater + Remi, traffic_test.py:27-64); sampled
    envs equal the oracle bit for bit after every call - ring indices, obs, rewards, waiting, every live car."""
    for var in ("TFX_RES_EPB", "TFX_RES_LPR", "TFX_RESIDENT", "TFX_RES_MIN_TICKS", "TFX_PAIRS", "TFX_LAYOUT",
                "TFX_MOVE_VARIANT"):
        monkeypatch.delenv(var, raising=False)
    c = wl.CONFIGS["cfg1"]
    eng = wl.setup_engine("cfg1")
    E, C, r, I = eng.E, eng.C, eng.r, eng.I
    assert (E, eng.R, C) == (1024, 80, 34) and eng.fused_ticks()[1]
    sample = [0, 1, 255, 256, 511, 777, 1023]
    orcs = [oracle_for(eng, c, 1, k) for k in sample]

    def check(tag):
        ld, lc = eng.leading.cpu().numpy(), eng.lastcar.cpu().numpy()
        ob, rw, wt = eng.obs.cpu().numpy(), eng.rewards.cpu().numpy(), eng.waiting.cpu().numpy()
        xs, vs = eng.x.cpu().numpy(), eng.v.cpu().numpy()
        for (orc, ids), k in zip(orcs, sample):
            assert np.array_equal(ld[k], orc.leading[0]) and np.array_equal(lc[k], orc.lastcar[0]), (tag, k)
            assert np.array_equal(ob[k], orc.obs[0]) and np.array_equal(wt[k], orc.waiting[0]), (tag, k)
            assert np.array_equal(rw[k], orc.rewards[0]), (tag, k)
            live = live_mask(ld[k], lc[k], C)
            assert np.array_equal(xs[k][live].view(np.int32), orc.x[0][live].view(np.int32)), (tag, k)
            assert np.array_equal(vs[k][live].view(np.int32), orc.v[0][live].view(np.int32)), (tag, k)

    t = 0
    for chunk in [1, 7, 10, 3, 9, 13, 17]:
        eng.step(chunk)
        assert eng.step_kernel() == "k_res"
        for _ in range(chunk):
            for (orc, ids) in orcs:
                step_oracle(orc, ids, eng, t, threads=1)
            t += 1
        check(t)
    assert t == 60 and eng.fused_ticks()[0] == 60
    # two fused decisions: obs accumulation, remi reward and `if done: break` per env (traffic_test.py:37-64)
    for dec in range(2):
        aobs, arew, adone = eng.agent_step(10, remi=True)
        aobs, arew, adone = aobs.cpu().numpy(), arew.cpu().numpy(), adone.cpu().numpy()
        for (orc, ids), k in zip(orcs, sample):
            total_obs, done = np.zeros(2 * r + I, np.float32), False
            for j in range(10):
                obs, rew, d = step_oracle(orc, ids, eng, t + j, threads=1)
                obs, done = obs[0], bool(d[0])
                total_obs[:r] += obs[:r]
                total_obs[r:2 * r] = obs[r:2 * r]
                total_obs[-I:] = obs[-I:] / 100 * (2 * obs[-2 * I:-I] - 1)
                if done:
                    break
            assert not done        # (the benchmark's traffic does not overflow this early: the clocks stay aligned)
            assert np.array_equal(aobs[k], total_obs), (dec, k)
            assert np.array_equal(arew[k], orc.remi_reward()[0]), (dec, k)
            assert adone[k] == 0, (dec, k)
        t += 10
        ld, lc = eng.leading.cpu().numpy(), eng.lastcar.cpu().numpy()
        for (orc, ids), k in zip(orcs, sample):
            assert np.array_equal(ld[k], orc.leading[0]) and np.array_equal(lc[k], orc.lastcar[0]), (dec, k)
            assert np.array_equal(eng.waiting[k].cpu().numpy(), orc.waiting[0]), (dec, k)
    assert eng.step_kernel() == "k_res"


@pytest.mark.parametrize("envs,expect", [(600, "split"), (300, "tail"), (96, "pairs"), (40, "graph"), (6, "small")])
def test_default_heuristics_at_mid_batches_vs_oracle(envs, expect, monkeypatch):
    """The handle's OWN choices (no TFX_* switch set) between the headline batch and the tiny ones the rest of the suite
    forces paths on: 600 envs of the 16x16 grid take pairs + k_tail in two halves on two streams, 300 pairs + k_tail in
    one range, 96 pairs with the three per-road launches and every tile's walk split over two wavefronts (k_move_tts), 40
    and 6 the same with four / eight wavefronts per tile - the launch-bound ones replayed as a HIP graph; a call's odd
    last tick on k_move_ts where the launch is small enough for it.  Calls of even and odd lengths and fused decisions in between; sampled envs equal the oracle bit for
    bit, and the counters say the expected path ran."""
    for var in ("TFX_RES_EPB", "TFX_RES_LPR", "TFX_RESIDENT", "TFX_RES_MIN_TICKS", "TFX_PAIRS", "TFX_LAYOUT", "TFX_TAIL",
                "TFX_SPLIT", "TFX_MOVE_VARIANT", "TFX_GRAPH", "TFX_KINDS", "TFX_FASTDIV", "TFX_TT_SEG"):
        monkeypatch.delenv(var, raising=False)
    c = wl.CONFIGS["cfg2"]
    eng = wl.setup_engine("cfg2", envs=envs)
    C = eng.C
    sample = [0, envs // 2 - 1, envs // 2, envs - 1]
    orcs = [oracle_for(eng, c, 1, k) for k in sample]
    t = 0
    for chunk in [2, 7, 10, 1, 6, 5]:
        eng.step(chunk)
        for _ in range(chunk):
            for (orc, ids) in orcs:
                step_oracle(orc, ids, eng, t, threads=1)
            t += 1
        ld, lc = eng.leading.cpu().numpy(), eng.lastcar.cpu().numpy()
        for (orc, ids), k in zip(orcs, sample):
            assert np.array_equal(ld[k], orc.leading[0]) and np.array_equal(lc[k], orc.lastcar[0]), (t, k)
            assert np.array_equal(eng.obs[k].cpu().numpy(), orc.obs[0]), (t, k)
            assert np.array_equal(eng.waiting[k].cpu().numpy(), orc.waiting[0]), (t, k)
            assert np.array_equal(eng.rewards[k].cpu().numpy(), orc.rewards[0]), (t, k)
    for (orc, ids), k in zip(orcs, sample):
        live = live_mask(ld[k], lc[k], C)
        xk, vk = eng.x[k].cpu().numpy(), eng.v[k].cpu().numpy()
        assert np.array_equal(xk[live].view(np.int32), orc.x[0][live].view(np.int32)), k
        assert np.array_equal(vk[live].view(np.int32), orc.v[0][live].view(np.int32)), k
    pairs = 2 + 6 + 10 + 6 + 4
    if expect == "split":
        assert eng.pair_ticks() == pairs and eng.tail_ticks() == pairs and eng.split_ticks() == 2 + 7 + 10 + 6 + 5
    elif expect == "tail":
        assert eng.pair_ticks() == pairs and eng.tail_ticks() == pairs and eng.split_ticks() == 0
    else:
        assert eng.pair_ticks() == pairs and eng.tail_ticks() == 0 and eng.split_ticks() == 0
        # (the last call's last tick was a single one: k_move_ts where the launch is small enough for it - 512 tiles)
        assert eng.step_kernel() == ("k_move_ts" if envs * 17 <= 512 else "k_move_tt")
    # a fused decision on the same handle, against a second handle that takes it tick by tick
    ref = wl.setup_engine("cfg2", envs=envs)
    ref.step(31)
    monkeypatch.setenv("TFX_PAIRS", "0")
    slow = wl.setup_engine("cfg2", envs=envs)
    monkeypatch.delenv("TFX_PAIRS")
    slow.step(31)
    for _ in range(2):
        ra = [x.clone() for x in ref.agent_step(10, remi=True)]
        rb = [x.clone() for x in slow.agent_step(10, remi=True)]
        for x, y in zip(ra, rb):
            assert torch.equal(x, y)
    assert torch.equal(ref.leading, slow.leading) and torch.equal(ref.obs, slow.obs) and torch.equal(ref.waiting, slow.waiting)
