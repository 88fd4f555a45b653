"""Heterogeneous cars: the reference's `archetypes` is a TABLE and add_new_cars draws a row per car
(traffic_env.py:35-43,164); every car keeps its parameters (length, accelerations, desired speed, headway, gap,
exponent) through every handoff.  Runs captured from the reference with three / two rows (one with delta = 2), replayed
through tfx_step against the oracle (bit-exact, whole run), against the reference's integers (first 120 ticks) and
teacher-forced against its floats (conftest.assert_floats_match_reference); random pathological ring states of mixed rows against the oracle."""
import numpy as np
import pytest

from conftest import assert_floats_match_reference, golden_names
from oracle.oracle import OracleEnv, live_mask
from test_gpu_parity import counts, same_bits

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from gym_traffic.core import TfxEngine  # noqa: E402
from gym_traffic._native import TfxError  # noqa: E402

ROWS = [1, 2, 3, 4, 5, 6, 7, 8]      # v, l, a, delta, v0, b, T, s0 of the reference's 10-column row


def table8(tab10):
    return np.asarray(tab10, np.float32)[:, ROWS]


def engine(g, n_envs=1):
    sc = g.sc
    return TfxEngine(sc["m"], sc["n"], sc["L"], sc["C"], n_envs=n_envs, rate=sc["rate"], planes=3,
                     archetypes=table8(g.archetypes))


def spawn_rows(eng, roads, rows):
    """uint8 [1, n_entry, S]: row of the j-th car each entry road receives, in creation order."""
    per = {}
    for rd, a in zip(roads, rows):
        per.setdefault(int(rd), []).append(int(a))
    S = max([len(v) for v in per.values()] + [1])
    out = np.zeros((1, max(1, eng.n_entry), S), np.uint8)
    for rd, v in per.items():
        out[0, eng.entry_index[rd], :len(v)] = v
    return out


def assert_cars_equal(eng, orc, tab10, where):
    ld, lc = eng.leading.cpu().numpy(), eng.lastcar.cpu().numpy()
    assert np.array_equal(ld, orc.leading) and np.array_equal(lc, orc.lastcar), where
    assert np.array_equal(eng.obs.cpu().numpy(), orc.obs), where
    assert np.array_equal(eng.rewards.cpu().numpy(), orc.rewards), where
    assert np.array_equal(eng.waiting.cpu().numpy(), orc.waiting), where
    x, v, w = eng.planes_numpy()
    a = eng.arch.cpu().numpy()
    for k in range(eng.E):
        live = live_mask(ld[k], lc[k], eng.C)
        assert same_bits(x[k][live], orc.x[k][live]) and same_bits(v[k][live], orc.v[k][live]), where
        assert np.array_equal(w[k][live], orc.w[k][live]), where
        assert np.array_equal(a[k][live], orc.arch_plane(k, tab10)[live]), where


@pytest.mark.parametrize("name", golden_names(True))
def test_archetype_runs_vs_oracle_and_reference(name, golden_cache):
    g = golden_cache(name)
    sc, tab = g.sc, g.archetypes
    eng = engine(g)
    orc = OracleEnv(sc["m"], sc["n"], sc["L"], sc["C"], g["dest"], g["phases"], g["nexts"], rate=sc["rate"])
    eng.reset(g["init_phase"])
    orc.reset(g["init_phase"])
    seen = set()
    for t in range(sc["T"]):
        roads, rows = g.spawns(t), g.spawn_archs(t)
        eng.set_spawns(counts=counts(eng, [roads]), rows=spawn_rows(eng, roads, rows))
        eng.set_actions(g["actions"][t][None, :])
        eng.step(1)
        _, _, od = orc.step(g["actions"][t], [roads], spawn_arch=[rows], archetypes=tab)
        k = t + 1
        assert np.array_equal(eng.done.cpu().numpy(), od), (name, k)
        if k % 5 == 0 or k < 25 or k == sc["T"]:
            assert_cars_equal(eng, orc, tab, "%s tick %d" % (name, k))
        if k <= 120:
            assert np.array_equal(eng.leading[0].cpu().numpy(), g["leading"][k]), (name, k)
            assert np.array_equal(eng.lastcar[0].cpu().numpy(), g["lastcar"][k]), (name, k)
            assert np.array_equal(eng.obs[0].cpu().numpy(), g["obs"][k]), (name, k)
            assert np.array_equal(eng.rewards[0].cpu().numpy(), g["rewards"][k]), (name, k)
        seen |= set(rows.tolist())
    assert seen == set(range(len(tab))) and eng.step_kernel() == "k_move_t"
    live = live_mask(eng.leading[0].cpu().numpy(), eng.lastcar[0].cpu().numpy(), sc["C"])
    assert len(set(eng.arch[0].cpu().numpy()[live].tolist())) == len(tab)      # every row is on the roads at the end


@pytest.mark.parametrize("name", golden_names(True))
def test_archetype_runs_teacher_forced_vs_reference(name, golden_cache):
    g = golden_cache(name)
    sc, tab = g.sc, g.archetypes
    eng = engine(g)
    for t in range(0, sc["T"], 3):
        eng.load_state(g["state_x"][t][None], g["state_v"][t][None], g["leading"][t][None], g["lastcar"][t][None],
                       w=g["state_w"][t][None], arch=g["state_a"][t][None])
        eng.obs.copy_(torch.as_tensor(g["obs"][t][None]))
        eng.rewards.copy_(torch.as_tensor(g["rewards"][t][None]))
        eng.waiting.copy_(torch.as_tensor(g["waiting"][t][None]))
        eng.passed_dst.copy_(torch.as_tensor(g["passed_dst"][t][None]))
        if t > 0 and t % sc["remi_every"] == 0:
            eng.remi_reward()
        eng.set_tick(t)
        roads, rows = g.spawns(t), g.spawn_archs(t)
        eng.set_spawns(counts=counts(eng, [roads]), rows=spawn_rows(eng, roads, rows))
        eng.set_actions(g["actions"][t][None, :])
        eng.step(1)
        k = t + 1
        ld, lc = eng.leading[0].cpu().numpy(), eng.lastcar[0].cpu().numpy()
        assert np.array_equal(ld, g["leading"][k]) and np.array_equal(lc, g["lastcar"][k]), (name, k)
        assert np.array_equal(eng.obs[0].cpu().numpy(), g["obs"][k]), (name, k)
        assert np.array_equal(eng.waiting[0].cpu().numpy(), g["waiting"][k]), (name, k)
        x, v, w = [p[0] for p in eng.planes_numpy()]
        live = live_mask(ld, lc, sc["C"])
        if live.any():
            assert_floats_match_reference(x[live], v[live], g["state_x"][k][live], g["state_v"][k][live],
                                          a_max=float(tab[:, 3].max()), rate=sc["rate"], where=(name, k))
            assert np.array_equal(w[live], g["state_w"][k][live]), (name, k)
            assert np.array_equal(eng.arch[0].cpu().numpy()[live], g["state_a"][k][live]), (name, k)


@pytest.mark.parametrize("C,validate", [(10, False), (34, True), (130, False)])
def test_mixed_rows_on_pathological_states_vs_oracle(C, validate):
    """Random ring states (wrapped, full, empty roads, several cars past the road end so that handed-off cars cascade
    and envs take the serial advance, unsorted cars), every car with a random row of a four-row table whose exponents
    are 1, 2, 4, 8, 2.5 and 0.75; bursts of arrivals of mixed rows; validate mode carries the spawn ticks beside the rows."""
    from test_gpu_parity import random_state
    tab8 = MIXED_ROWS
    n_rows = len(tab8)
    tab10 = np.zeros((n_rows, 10), np.float32)
    tab10[:, ROWS] = tab8
    m, n, L, E = 3, 2, 150.0, 4
    eng = TfxEngine(m, n, L, C, n_envs=E, planes=3, validate=validate, archetypes=tab8)
    orc = OracleEnv(m, n, L, C, eng.dest, eng.phases, eng.nexts, n_envs=E, validate=validate)
    rng = np.random.RandomState(900 + C)
    serial_ticks = 0
    for trial in range(5):
        x, v, w, leading, lastcar = random_state(rng, E, eng.R, C, L, crowd=rng.choice([0.3, 0.8]),
                                                 beyond=rng.choice([0.0, 0.05, 0.4, 1.6]), sorted_x=bool(trial % 2))
        arch = rng.randint(0, n_rows, size=x.shape).astype(np.uint8)
        phase = rng.randint(2, size=(E, eng.I)).astype(np.int32)
        eng.reset(phase)
        orc.reset(phase)
        eng.load_state(x, v, leading, lastcar, w=w, arch=arch)
        for k in range(E):
            orc.load_planes(k, x[k], v[k], w[k], leading[k], lastcar[k], arch=arch[k], archetypes=tab10)
        eng.set_tick(60)
        orc.steps[:] = 60
        for t in range(8):
            act = rng.randint(2, size=(E, eng.I)).astype(np.int32)
            roads = [rng.choice(eng.entrypoints, size=rng.randint(0, 5)).tolist() for _ in range(E)]
            rows = [rng.randint(0, n_rows, size=len(r)).tolist() for r in roads]
            per_env = [spawn_rows(eng, r, a)[0] for r, a in zip(roads, rows)]
            S = max(p.shape[-1] for p in per_env)
            buf = np.zeros((E, max(1, eng.n_entry), S), np.uint8)
            for k, p in enumerate(per_env):
                buf[k, :, :p.shape[-1]] = p
            eng.set_spawns(counts=counts(eng, roads), rows=buf)
            eng.set_actions(act)
            eng.step(1)
            _, _, od = orc.step(act, roads, spawn_arch=rows, archetypes=tab10)
            assert np.array_equal(eng.done.cpu().numpy(), od), (trial, t)
            assert_cars_equal(eng, orc, tab10, "C=%d trial %d tick %d" % (C, trial, t))
        if validate:
            nt = eng.n_trips.cpu().numpy()
            assert np.array_equal(nt, orc.n_trips)
            for k in range(E):
                assert np.array_equal(eng.trip_times[k, :nt[k]].cpu().numpy(), orc.trip_times[k, :nt[k]])


# exponents 4, 1, 2, 8 (multiply chains) and two that are no integers (include/tfx_pow.h)
MIXED_ROWS = np.array([[11.11, 4, 3, 4, 13.89, 6, 2, 1], [8.0, 8, 1.5, 1, 10.0, 4, 2.5, 2],
                      [12.0, 3.5, 4, 2, 16.0, 7, 1.5, 1], [9.0, 12, 1.0, 8, 11.0, 3, 3.0, 3],
                      [10.0, 5, 2.5, 2.5, 14.0, 5, 1.8, 1.5], [7.0, 6, 2.0, 0.75, 9.0, 4, 2.2, 2]], np.float32)


@pytest.mark.parametrize("tail,split", [("2", "0"), ("0", "0"), ("2", "2")])
@pytest.mark.parametrize("C,validate", [(10, False), (34, True), (66, False)])
def test_mixed_rows_in_two_tick_pairs_vs_oracle(C, validate, tail, split):
    """The same pathological states through multi-tick calls: the HET forms of the two-tick pass, of k_edge / k_tail and
    the two-stream split (forced at test size).  A car's own row drives BOTH of its ticks in the pass, its leader's row
    the gap; arrivals of mixed rows come from per-tick count and row buffers."""
    run_pairs_case(C, validate, tail, split, MIXED_ROWS, 60)


@pytest.mark.parametrize("tail,split", [("2", "2"), ("0", "0")])
def test_rows_survive_a_long_clock_and_a_wide_table(tail, split):
    """Round-3 advisor finding: the side word held 8 * tick + row as a FLOAT value, so from tick 2^21 on the row was
    rounded away and new cars got wrong parameters.  It now holds (tick mod 2^24) << 6 | row as integer bits: 40 table
    rows (the old limit was 8), the clock at 3 million ticks, validate mode - every car's row, spawn tick and trip time
    still equal the oracle's, which carries all ten parameters per car."""
    rng = np.random.RandomState(31)
    tab8 = np.stack([rng.uniform(7, 13, 40), rng.uniform(3, 12, 40), rng.uniform(1, 4, 40), rng.randint(1, 9, 40),
                     rng.uniform(9, 17, 40), rng.uniform(3, 7, 40), rng.uniform(1.2, 3, 40), rng.uniform(1, 3, 40)],
                    axis=1).astype(np.float32)
    run_pairs_case(34, True, tail, split, tab8, 3000000)


def run_pairs_case(C, validate, tail, split, tab8, tick0):
    from test_gpu_fused import engine_with
    from test_gpu_parity import random_state
    n_rows = len(tab8)
    tab10 = np.zeros((n_rows, 10), np.float32)
    tab10[:, ROWS] = tab8
    m, n, L, E = 3, 2, 150.0, 5
    eng = engine_with({"TFX_RESIDENT": "0", "TFX_PAIRS": "2", "TFX_TAIL": tail, "TFX_SPLIT": split}, E, planes=3,
                      m=m, n=n, length=L, capacity=C, validate=validate, archetypes=tab8)
    orc = OracleEnv(m, n, L, C, eng.dest, eng.phases, eng.nexts, n_envs=E, validate=validate)
    rng = np.random.RandomState(4100 + C)
    ran = 0
    for trial, T in enumerate([2, 5, 4, 7, 3, 6]):
        x, v, w, leading, lastcar = random_state(rng, E, eng.R, C, L, crowd=rng.choice([0.3, 0.8]),
                                                 beyond=rng.choice([0.0, 0.05, 0.4, 1.6]), sorted_x=bool(trial % 2))
        arch = rng.randint(0, n_rows, size=x.shape).astype(np.uint8)
        if tick0 > 400:     # the cars on the roads were spawned in the last few hundred ticks
            w = (tick0 - 400 + 8 * w).astype(np.float32)
        phase = rng.randint(2, size=(E, eng.I)).astype(np.int32)
        eng.reset(phase)
        orc.reset(phase)
        eng.load_state(x, v, leading, lastcar, w=w, arch=arch)
        for k in range(E):
            orc.load_planes(k, x[k], v[k], w[k], leading[k], lastcar[k], arch=arch[k], archetypes=tab10)
        eng.set_tick(tick0)
        orc.steps[:] = tick0
        acts = rng.randint(2, size=(T, E, eng.I)).astype(np.int32)
        roads = [[rng.choice(eng.entrypoints, size=rng.randint(0, 4)).tolist() for _ in range(E)] for _ in range(T)]
        rows = [[rng.randint(0, n_rows, size=len(r)).tolist() for r in rt] for rt in roads]
        S = 4
        buf = np.zeros((T, E, max(1, eng.n_entry), S), np.uint8)
        for t in range(T):
            for k in range(E):
                pr = spawn_rows(eng, roads[t][k], rows[t][k])[0]
                buf[t, k, :, :pr.shape[-1]] = pr
        eng.set_actions(acts, per_tick=True)
        eng.set_spawns(counts=np.stack([counts(eng, r) for r in roads]), per_tick=True, rows=buf)
        eng.step(T)
        ran += 2 * (T // 2)
        for t in range(T):
            orc.step(acts[t], roads[t], spawn_arch=rows[t], archetypes=tab10)
        assert_cars_equal(eng, orc, tab10, "C=%d trial %d (%d ticks)" % (C, trial, T))
        if validate:
            nt = eng.n_trips.cpu().numpy()
            assert np.array_equal(nt, orc.n_trips)
            for k in range(E):
                kk = min(int(nt[k]), eng.trip_cap)
                assert np.array_equal(eng.trip_times[k, :kk].cpu().numpy(), orc.trip_times[k, :kk])
    assert eng.pair_ticks() == ran and eng.step_kernel() == "k_move_tt"
    assert eng.tail_ticks() == (ran if tail == "2" else 0) and (eng.split_ticks() > 0) == (split == "2")


@pytest.mark.parametrize("remi", [False, True])
def test_agent_steps_of_mixed_rows_on_a_handle_that_runs_pairs(remi):
    """tfx_agent_step with heterogeneous cars in two-tick pairs (k_risk bounds a car's movement with the table's LARGEST
    acceleration), plain step() calls in between (they leave columns that start a row or two down).  Equal, bit for bit
    incl. every car's row, to a handle that never uses the pairs (k_move_t<HET> + k_advance), through decisions that
    overflow."""
    from test_gpu_fused import engine_with
    from test_gpu_parity import random_state
    tab8 = MIXED_ROWS
    m, n, L, C, E = 3, 3, 120.0, 14, 9
    kw = dict(planes=3, m=m, n=n, length=L, capacity=C, archetypes=tab8)
    a = engine_with({"TFX_RESIDENT": "0", "TFX_PAIRS": "2", "TFX_TAIL": "2", "TFX_SPLIT": "2"}, E, **kw)
    c = engine_with({"TFX_RESIDENT": "0", "TFX_PAIRS": "0"}, E, **kw)
    rng = np.random.RandomState(777 + int(remi))
    froze = 0
    for trial, T in enumerate([3, 10, 4, 6]):
        x, v, w, leading, lastcar = random_state(rng, E, a.R, C, L, crowd=rng.choice([0.5, 0.9]),
                                                 beyond=rng.choice([0.0, 0.05, 0.4]), sorted_x=bool(trial % 2))
        arch = rng.randint(0, len(tab8), size=x.shape).astype(np.uint8)
        phase = rng.randint(2, size=(E, a.I)).astype(np.int32)
        act = rng.randint(2, size=(E, a.I)).astype(np.int32)
        period = int(rng.choice([1, 2, 5]))
        for eng in (a, c):
            eng.reset(phase)
            eng.load_state(x, v, leading, lastcar, w=w, arch=arch)
            eng.set_tick(40)
            eng.set_spawns(period=period)
            eng.set_actions(act)
        for step in range(3):
            p0 = a.pair_ticks()
            ra = [t.clone() for t in a.agent_step(T, remi=remi)]
            rc = [t.clone() for t in c.agent_step(T, remi=remi)]
            assert a.pair_ticks() == p0 + 2 * (T // 2)        # the decision itself ran in pairs
            for u, w_ in zip(ra, rc):
                assert torch.equal(u, w_), (trial, step)
            froze += int(ra[2].sum())
            a.step(3)                 # a pair and a single tick: leaves columns that start a row or two down
            c.step(3)
            for name in ("leading", "lastcar", "obs", "rewards", "waiting", "passed_dst", "done"):
                assert torch.equal(getattr(a, name), getattr(c, name)), (name, trial, step)
            ld, lc = a.leading.cpu().numpy(), a.lastcar.cpu().numpy()
            pa, pb = a.planes_numpy(), c.planes_numpy()
            aa, ab = a.arch.cpu().numpy(), c.arch.cpu().numpy()
            for k in range(E):
                live = live_mask(ld[k], lc[k], C)
                for u, w_ in zip(pa, pb):
                    assert same_bits(u[k][live], w_[k][live]), (trial, step, k)
                assert np.array_equal(aa[k][live], ab[k][live]), (trial, step, k)
    assert a.pair_ticks() > 0 and c.pair_ticks() == 0 and froze > 0


def test_unsupported_archetype_tables_are_refused():
    with pytest.raises(TfxError):          # an exponent outside (0, 64]
        TfxEngine(2, 2, 100.0, 10, planes=3, archetypes=[[11.11, 4, 3, -1.0, 13.89, 6, 2, 1]])
    frac = TfxEngine(2, 2, 100.0, 10, planes=3, archetypes=[[11.11, 4, 3, 2.5, 13.89, 6, 2, 1]])   # any other: per-car path
    assert frac.het
    with pytest.raises(TfxError):          # several rows need the side word (planes = 3)
        TfxEngine(2, 2, 100.0, 10, planes=2, archetypes=[[11.11, 4, 3, 4, 13.89, 6, 2, 1], [8, 8, 1.5, 4, 10, 4, 2.5, 2]])
    eng = TfxEngine(2, 2, 100.0, 10, planes=2, archetypes=[[9.0, 5, 2, 4, 12.0, 5, 2, 1.5]])   # ONE ordinary row: fine
    assert not eng.het


@pytest.mark.parametrize("name", golden_names(True))
def test_gym_surface_with_a_user_archetype_table(name, golden_cache):
    """What a user of the reference does: replace the module's `archetypes` table, make the env, seed it, step.  The
    spawner replays the reference's RandomState draws (exponential, randint(rows), choice), so the same cars of the
    same rows arrive on the same roads; 120 ticks of integers equal the reference's, env.state shows every car's
    ten parameters at the reference's indices, and the fused Repeater path (env.repeat) agrees with tick-by-tick."""
    import gym_traffic  # noqa: F401
    import gym
    from gym_traffic.envs import traffic_env as te
    from gym_traffic.envs.roadgraph import GridRoad
    from gym_traffic.flags import update_flags
    g = golden_cache(name)
    sc, tab = g.sc, g.archetypes
    keep = te.archetypes
    try:
        te.archetypes = tab.copy()
        update_flags(poisson=bool(sc["poisson"]), rate=float(sc["rate"]), local_cars_per_sec=float(sc["lcps"]),
                     entry=sc["entry"], learn_switch=bool(sc["learn_switch"]), mode=sc["mode"])

        def make():
            env = gym.make('traffic-v0')
            env.set_graph(GridRoad(sc["m"], sc["n"], sc["L"]), capacity=sc["C"])
            env.seed_generator(sc["seed"])
            env.reset_entrypoints()
            np.random.seed(sc["seed"])
            env.reset()
            return env
        env = make()
        assert env.engine.het and np.array_equal(env.current_phase, g["init_phase"])
        for t in range(120):
            obs, rew, done, _ = env.step(g["actions"][t])
            k = t + 1
            assert np.array_equal(obs, g["obs"][k]) and np.array_equal(rew, g["rewards"][k]), (name, k)
            assert bool(done) == bool(g["done"][k]), (name, k)
            assert np.array_equal(np.asarray(env.leading), g["leading"][k]), (name, k)
            assert np.array_equal(np.asarray(env.lastcar), g["lastcar"][k]), (name, k)
        assert env.generated_cars == int(g["spawn_off"][120])
        st = env.state.numpy()
        live = live_mask(g["leading"][120], g["lastcar"][120], sc["C"])
        rows = g["state_a"][120]
        for col in (te.li, te.ai, te.deltai, te.v0i, te.bi, te.ti, te.s0i):
            assert np.array_equal(st[:, col, :][live], tab[rows[live], col]), col
        assert np.array_equal(st[:, te.wi, :][live], g["state_w"][120][live])
        # the fused decision (what Repeater(10) submits) against ten single steps, from a fresh env each
        a, b = make(), make()
        for dec in range(6):
            act = g["actions"][10 * dec]
            tot, rsum, done_a = a.repeat(act, 10)
            want = np.zeros_like(tot)
            r = sc["m"] * sc["n"] * 4
            for _ in range(10):
                obs, rew, done_b, _ = b.step(act)
                want[:r] += obs[:r]
                want[r:2 * r] = obs[r:2 * r]
                want[2 * r:] = obs[-(len(obs) - 2 * r) // 2:] / 100 * (2 * obs[2 * r:2 * r + (len(obs) - 2 * r) // 2] - 1)
                if done_b:
                    break
            assert np.array_equal(tot, want) and bool(done_a) == bool(done_b), dec
            assert np.array_equal(np.asarray(a.leading), np.asarray(b.leading)), dec
            sa, sb = a.state.numpy(), b.state.numpy()
            lv = live_mask(np.asarray(a.leading), np.asarray(a.lastcar), sc["C"])
            assert np.array_equal(sa[:, :, :][:, :, :].transpose(0, 2, 1)[lv], sb.transpose(0, 2, 1)[lv]), dec
    finally:
        te.archetypes = keep
        update_flags(poisson=True, rate=0.5, local_cars_per_sec=0.12, entry='all', learn_switch=False, mode='train')
