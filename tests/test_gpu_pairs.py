"""Two ticks per pass over the cars (csrc/tfx_move_tt.hpp): k_move_tt takes every car but the head of its
road through two ticks in one trip through HBM, k_edge finishes the second tick for the heads and for the
cars that joined a road in between.  tfx_step uses the pairs on its own for big launches; here they are
forced at test sizes (TFX_PAIRS=2) and must be bit-identical to the tick-by-tick kernels and to the oracle:
pathological ring states (wrapped, full, empty, unsorted, cars several road lengths past the end so that
handed-off cars cascade, more than two pops per road and tick), per-tick action and spawn buffers, the
on-device rules, every capacity class, odd and even call lengths."""
import numpy as np
import pytest

from test_gpu_parity import (assert_engines_equal, assert_same_state, counts, load_both, oracle_like,
                             random_state)
from test_gpu_fused import engine_with

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from gym_traffic import workload as wl  # noqa: E402


_TAIL = ["2", "0", "0", "0"]


@pytest.fixture(params=["tail", "launches", "split", "seg_tail", "seg_launches", "seg4_tail", "seg8_launches"], autouse=True)
def pair_tail(request):
    """What follows a pass: k_tail - advance(t), the edge work of t+1 and advance(t+1) in one launch, a workgroup per
    env (csrc/tfx_tail.hpp, forced at test sizes) - or the three separate launches; "split": k_tail, and tfx_step runs
    the env range as two halves on two streams (tfx_split_ticks).  "seg_*": the pass with every tile's walk split over
    two, four or eight wavefronts (k_move_tts, csrc/tfx_move_tts.hpp - what launches that cannot fill the chip take),
    forced wherever that form exists (single-archetype cars, with or without the side-word plane)."""
    _TAIL[:] = ["0" if request.param.endswith("launches") else "2", "2" if request.param == "split" else "0",
                "2" if request.param.startswith("seg") else "0",
                "4" if request.param.startswith("seg4") else ("8" if request.param.startswith("seg8") else "2")]
    yield request.param
    _TAIL[:] = ["2", "0", "0", "0"]


def pairs_engine(E, **cfg):
    eng = engine_with({"TFX_RESIDENT": "0", "TFX_PAIRS": "2", "TFX_TAIL": _TAIL[0], "TFX_SPLIT": _TAIL[1], "TFX_TT_SEG": _TAIL[2], "TFX_TT_SEGS": _TAIL[3]}, E, **cfg)
    assert eng.pair_ticks() == 0
    return eng


def pertick_engine(E, **cfg):
    return engine_with({"TFX_RESIDENT": "0", "TFX_PAIRS": "0"}, E, **cfg)


def paired(T):
    """ticks of a T-tick call that run as pairs: all of them, or all but the last"""
    return 2 * (T // 2)


@pytest.mark.parametrize("m,n,C,length,validate", [(2, 2, 10, 60.0, False), (3, 2, 20, 120.0, True), (4, 4, 34, 200.0, False),
                                                   (2, 3, 66, 400.0, False), (5, 3, 12, 80.0, True), (1, 1, 6, 50.0, True),
                                                   (2, 2, 130, 800.0, False), (3, 3, 34, 150.0, True)])
@pytest.mark.parametrize("sorted_x", [True, False])
def test_pairs_random_states_vs_oracle(m, n, C, length, validate, sorted_x):
    """validate: the cars' spawn ticks travel through the pairs (the W forms of the pass, k_edge and k_tail) and the trip
    times of cars leaving the map (advance_hack, traffic_env.py:139-157) come out the same"""
    rng = np.random.RandomState(8642 + C + int(sorted_x))
    E = 5
    eng = pairs_engine(E, m=m, n=n, length=length, capacity=C, rate=0.5, validate=validate, planes=3 if validate else 2)
    orc = oracle_like(eng)
    ran = 0
    for trial, T in enumerate([3, 4, 7, 2, 5, 11, 1, 6]):
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
        ran += paired(T)
        done = np.zeros(E, bool)
        for t in range(T):
            done |= orc.step(acts[t], roads[t])[2].astype(bool)
        assert np.array_equal(eng.done.cpu().numpy().astype(bool), done), trial
        assert_same_state(eng, orc, "trial %d (%d ticks)" % (trial, T))
        if validate:
            nt = eng.n_trips.cpu().numpy()
            assert np.array_equal(nt, orc.n_trips), trial
            for k in range(E):
                kk = min(int(nt[k]), eng.trip_cap)
                assert np.array_equal(eng.trip_times[k, :kk].cpu().numpy(), orc.trip_times[k, :kk]), (trial, k)
    # (the last call had an even number of ticks: its last mover was a pass - in its several-wavefronts-per-tile form
    # where that was asked for)
    assert eng.pair_ticks() == ran and eng.step_kernel() == ("k_move_tts" if _TAIL[2] == "2" else "k_move_tt")
    assert eng.tail_ticks() == (ran if _TAIL[0] == "2" else 0)
    assert (eng.split_ticks() > 0) == (_TAIL[1] == "2")


@pytest.mark.parametrize("chunk", [3, 10, 25])
def test_pairs_equal_tick_by_tick_on_device_rules(chunk):
    """The bench's inputs (fixed-cycle lights, periodic arrivals), dense enough that rings overflow: 60 ticks in
    calls of `chunk` ticks == the same with the pairs disabled; counters and done flags included."""
    E, T = 9, 60
    cfg = dict(m=4, n=4, length=200.0, capacity=34, rate=0.5)
    a = pairs_engine(E, **cfg)
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
        a.step(min(chunk, T - done))
        done += min(chunk, T - done)
    c.step(T)
    assert a.pair_ticks() > 0 and c.pair_ticks() == 0
    assert_engines_equal(a, c)
    assert a.vehicle_updates() == c.vehicle_updates() > 0
    assert a.tick == c.tick == T
    assert torch.equal(a.done_tick, c.done_tick) and int(a.done_tick.max()) > 0


def test_pairs_golden_ints(golden_cache):
    """A captured reference run fed through per-tick buffers in uneven chunks: the integers equal the
    reference's for the first 120 ticks, everything equals the oracle."""
    g = golden_cache("g3x3_default")
    sc = g.sc
    eng = pairs_engine(1, m=sc["m"], n=sc["n"], length=sc["L"], capacity=sc["C"], rate=sc["rate"])
    orc = oracle_like(eng)
    eng.reset(g["init_phase"])
    orc.reset(g["init_phase"])
    t = 0
    for chunk in [3, 4, 5, 13, 10, 10, 7, 25, 3, 8, 32]:
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
    assert t == 120 and eng.pair_ticks() > 80


def test_pairs_golden_validate_trip_times(golden_cache):
    """The reference's validate-mode run (advance_hack's trip times, traffic_env.py:139-157) through the W forms of the
    pairs: the trip log equals the oracle's for the whole 400-tick run, and the reference's own over the first 120 ticks."""
    g = golden_cache("g2x2_validate")
    sc = g.sc
    eng = pairs_engine(1, m=sc["m"], n=sc["n"], length=sc["L"], capacity=sc["C"], rate=sc["rate"], validate=True, planes=3)
    orc = oracle_like(eng)
    eng.reset(g["init_phase"])
    orc.reset(g["init_phase"])
    T = len(g["actions"])
    t = 0
    for chunk in [2, 9, 4, 16, 3, 10, 10, 7, 25, 6, 8, 20, 51, 50, 50, 50, 50, 29]:
        chunk = min(chunk, T - t)
        if chunk <= 0:
            break
        acts = g["actions"][t:t + chunk][:, None, :]
        sp = np.stack([counts(eng, [g.spawns(t + j)]) for j in range(chunk)])
        eng.set_actions(acts, per_tick=True)
        eng.set_spawns(counts=sp, per_tick=True)
        eng.step(chunk)
        for j in range(chunk):
            orc.step(g["actions"][t + j], [g.spawns(t + j)])
        t += chunk
        assert_same_state(eng, orc, "tick %d" % t)
        if t == 120:        # (the horizon over which every fixture's integers are held to the reference's)
            n = int(g["trip_count"][120])
            assert int(eng.n_trips[0]) == n
            assert np.array_equal(eng.trip_times[0, :n].cpu().numpy().astype(np.float64), g["trip_times"][:n])
    n = int(eng.n_trips[0])
    assert t == T and n == int(orc.n_trips[0]) and n > 10 and eng.pair_ticks() > 300
    assert np.array_equal(eng.trip_times[0, :n].cpu().numpy(), orc.trip_times[0, :n])


def test_pairs_with_device_poisson_and_greedy():
    """On-device Poisson arrivals and the greedy controller produce the second tick's inputs between the two
    halves of a pair: same streams, same lights, same cars as tick by tick."""
    E = 6
    cfg = dict(m=4, n=3, length=150.0, capacity=20, rate=0.5)
    a = pairs_engine(E, **cfg)
    c = pertick_engine(E, **cfg)
    for eng in (a, c):
        eng.reset(np.zeros((E, eng.I), np.int32))
        eng.set_poisson(0.9, seed=99)
        eng.set_greedy(3)
        eng.reset_counters()
    for T in (3, 10, 4, 25, 7):
        a.step(T)
        c.step(T)
        assert_engines_equal(a, c)
    assert a.pair_ticks() > 30 and a.vehicle_updates() == c.vehicle_updates() > 0


@pytest.mark.parametrize("C", [6, 20, 34])
def test_pairs_full_rings_pop_and_receive_in_the_second_tick(C):
    """Every ring full, every head a few metres short of the road end: heads leave in the SECOND tick of a pair
    (k_edge leaves their columns starting one row down) while their predecessors hand cars over in the same
    tick - the append must not run past the tile's last row (found by the fuzzer, seed 32 case 706: it landed
    in the next tile's row 0)."""
    rng = np.random.RandomState(3200 + C)
    E, m, n, length = 6, 4, 4, 250.0
    eng = pairs_engine(E, m=m, n=n, length=length, capacity=C, rate=1.0)
    orc = oracle_like(eng)
    left_in_second = 0
    for trial, T in enumerate([3, 4, 3, 6, 5]):
        x = np.zeros((E, eng.R, C), np.float32)
        v = np.zeros((E, eng.R, C), np.float32)
        w = np.zeros((E, eng.R, C), np.float32)
        leading = rng.randint(1, C, size=(E, eng.R)).astype(np.int32)
        lastcar = leading.copy()
        for k in range(E):
            for e in range(eng.R):
                cnt = C - 2 if rng.rand() < 0.8 else C - 3
                head = length - rng.uniform(1.0, 14.0)
                s = int(leading[k, e])
                for j in range(cnt):
                    s = s + 1 if s + 1 < C else 1
                    x[k, e, s] = head - 5.5 * j
                    v[k, e, s] = rng.uniform(3.0, 13.0)
                lastcar[k, e] = s
                x[k, e, leading[k, e]] = np.inf
        phase = rng.randint(2, size=(E, eng.I)).astype(np.int32)
        elapsed = rng.randint(6, 12, size=(E, eng.I)).astype(np.int32)      # (past yellow: half the lights are green)
        load_both(eng, orc, x, v, w, leading, lastcar, phase, elapsed)
        eng.set_tick(60)
        orc.steps[:] = 60
        acts = np.repeat(phase[None], T, 0)
        roads = [[rng.choice(eng.entrypoints, size=rng.randint(0, 3)).tolist() for _ in range(E)] for _ in range(T)]
        eng.set_actions(acts, per_tick=True)
        eng.set_spawns(counts=np.stack([counts(eng, r) for r in roads]), per_tick=True)
        eng.step(T)
        for t in range(T):
            orc.step(acts[t], roads[t])
            if t % 2 == 1 and t < paired(T):                     # second tick of a pair
                left_in_second += int(orc.obs[:, :eng.r].sum())
        assert_same_state(eng, orc, "C=%d trial %d (%d ticks)" % (C, trial, T))
    assert eng.pair_ticks() > 0 and left_in_second > 0


@pytest.mark.parametrize("m,n,C,length,validate", [(2, 2, 10, 60.0, False), (4, 4, 20, 250.0, True), (3, 2, 34, 200.0, False),
                                                   (2, 2, 130, 800.0, False), (3, 3, 14, 90.0, True)])
@pytest.mark.parametrize("remi", [False, True])
def test_agent_steps_in_pairs_on_pathological_states(m, n, C, length, validate, remi):
    """tfx_agent_step over two-tick passes == tick by tick, from states where the first tick of a pair overflows
    rings, pops more than two cars per road or sends cars through a whole road (k_risk must sort those envs
    out: an env that overflows stands still for the rest of the step), and where the SECOND tick overflows (the
    env freezes with columns k_edge left uncompacted: the step's last launch moves them up)."""
    rng = np.random.RandomState(555 + C + int(remi))
    E = 12
    kw = dict(m=m, n=n, length=length, capacity=C, rate=0.5, validate=validate, planes=3 if validate else 2)
    a = pairs_engine(E, **kw)
    c = pertick_engine(E, **kw)
    froze = 0
    for trial, T in enumerate([3, 10, 4, 5, 10, 7]):
        x, v, w, leading, lastcar = random_state(rng, E, a.R, C, length, crowd=rng.choice([0.5, 0.9]),
                                                 beyond=rng.choice([0.0, 0.05, 0.4, 1.6]), sorted_x=bool(trial % 2))
        phase = rng.randint(2, size=(E, a.I)).astype(np.int32)
        act = rng.randint(2, size=(E, a.I)).astype(np.int32)
        period = int(rng.choice([1, 2, 5]))
        for eng in (a, c):
            eng.reset(phase)
            eng.load_state(x, v, leading, lastcar, w=w if validate else None)
            eng.set_tick(40)
            eng.set_spawns(period=period)
            eng.set_actions(act)
            if validate:
                eng.n_trips.zero_()
        for step in range(3):      # (an env that overflowed simply goes on in the next step, as a caller that does not reset it would)
            ra = [t.clone() for t in a.agent_step(T, remi=remi)]
            rc = [t.clone() for t in c.agent_step(T, remi=remi)]
            for u, w_ in zip(ra, rc):
                assert torch.equal(u, w_), (trial, step)
            assert_engines_equal(a, c)
            if validate:
                assert torch.equal(a.n_trips, c.n_trips), (trial, step)
                for k in range(E):
                    kk = min(int(a.n_trips[k]), a.trip_cap)
                    assert torch.equal(a.trip_times[k, :kk], c.trip_times[k, :kk]), (trial, step, k)
            froze += int(ra[2].sum())
    assert a.pair_ticks() > 0 and c.pair_ticks() == 0 and froze > 0


def test_columns_left_by_a_pair_between_calls():
    """A call may END on a pair, leaving columns whose live rows start one or two rows down.  Everything that
    touches the cars between calls must honour or clear that: the ring export behind eng.xv, masked episode
    resets, load_state, the two halves of a tick called on their own, single ticks, agent steps."""
    rng = np.random.RandomState(9090)
    E = 10
    cfg = dict(m=4, n=4, length=200.0, capacity=20, rate=1.0)
    a = pairs_engine(E, **cfg)
    c = pertick_engine(E, **cfg)
    x, v, leading, lastcar = wl.prefill_one_env(4, 4, 200.0, 20, 14, 9.0)
    for eng in (a, c):
        eng.reset(np.zeros((E, eng.I), np.int32))
        eng.load_state(np.repeat(x[None], E, 0), np.repeat(v[None], E, 0), np.repeat(leading[None], E, 0),
                       np.repeat(lastcar[None], E, 0))
        eng.set_spawns(period=2)
        eng.set_actions(cycle_period=5)
    def both(f):
        f(a)
        f(c)
        assert_engines_equal(a, c)
    both(lambda e: e.step(4))                      # ends on a pair
    both(lambda e: e.step(2))
    mask = rng.rand(E) < 0.4
    ph = rng.randint(2, size=(E, a.I)).astype(np.int32)
    both(lambda e: e.reset_envs(mask, ph))         # some envs start over, the others keep their columns
    both(lambda e: e.step(6))
    both(lambda e: (e.move_cars(), e.advance_finished_cars()))     # one tick as its two halves
    both(lambda e: e.step(1))
    both(lambda e: e.step(2))
    xs, vs, _ = a.planes_numpy()                   # ring image of the pair engine -> into both, shuffled by env
    perm = rng.permutation(E)
    ld, lc = a.leading.cpu().numpy()[perm], a.lastcar.cpu().numpy()[perm]
    both(lambda e: e.load_state(xs[perm], vs[perm], ld, lc))
    both(lambda e: e.step(8))
    for remi in (False, True):
        ra = [t.clone() for t in a.agent_step(4, remi=remi)]
        rc = [t.clone() for t in c.agent_step(4, remi=remi)]
        for u, w_ in zip(ra, rc):
            assert torch.equal(u, w_)
        assert_engines_equal(a, c)
    both(lambda e: e.step(3))
    assert a.pair_ticks() >= 4 + 2 + 6 + 2 + 8 + 2 and c.pair_ticks() == 0


def test_second_tick_pops_more_than_the_outbox_holds():
    """Unsorted columns: a head just short of the road end with cars BEYOND the end queued behind it (the pop prefix
    of :123 stops at the head, so they stay).  In the second tick of a pair the head leaves and takes the whole run
    with it: more pops than the outbox holds - k_edge leaves the road uncompacted, its env takes the serial
    advance, which meets the row offsets and full columns of the env's other roads."""
    rng = np.random.RandomState(4711)
    E, m, n, C, length = 8, 3, 3, 20, 250.0
    eng = pairs_engine(E, m=m, n=n, length=length, capacity=C, rate=0.5)
    orc = oracle_like(eng)
    big = 0
    for trial, T in enumerate([2, 4, 3, 6]):
        x = np.zeros((E, eng.R, C), np.float32)
        v = np.zeros((E, eng.R, C), np.float32)
        w = np.zeros((E, eng.R, C), np.float32)
        leading = rng.randint(1, C, size=(E, eng.R)).astype(np.int32)
        lastcar = leading.copy()
        for k in range(E):
            for e in range(eng.R):
                kind = rng.randint(3)
                cnt = C - 2 if kind == 0 else int(rng.randint(3, C - 2))
                s = int(leading[k, e])
                for j in range(cnt):
                    s = s + 1 if s + 1 < C else 1
                    if kind == 1 and j == 0:
                        x[k, e, s], v[k, e, s] = length - 3.0, 4.0          # leaves in the second tick
                    elif kind == 1 and j <= 4:
                        x[k, e, s], v[k, e, s] = length + 10.0 * j, 0.0     # already beyond the end, behind the head
                    else:
                        x[k, e, s] = length - 12.0 - 5.5 * j
                        v[k, e, s] = rng.uniform(0.0, 6.0)
                lastcar[k, e] = s
                x[k, e, leading[k, e]] = np.inf
        phase = rng.randint(2, size=(E, eng.I)).astype(np.int32)
        elapsed = rng.randint(6, 12, size=(E, eng.I)).astype(np.int32)
        load_both(eng, orc, x, v, w, leading, lastcar, phase, elapsed)
        eng.set_tick(10)
        orc.steps[:] = 10
        acts = np.repeat(phase[None], T, 0)
        roads = [[[] for _ in range(E)] for _ in range(T)]
        eng.set_actions(acts, per_tick=True)
        eng.set_spawns(counts=np.stack([counts(eng, r) for r in roads]), per_tick=True)
        eng.step(T)
        for t in range(T):
            orc.step(acts[t], roads[t])
            if t == 1:
                big = max(big, int(orc.obs[:, :eng.r].max()))
        assert_same_state(eng, orc, "trial %d (%d ticks)" % (trial, T))
    assert big > 2 and eng.pair_ticks() > 0


def test_stale_risk_stamps_after_the_clock_is_set_back():
    """k_risk stamps risky envs with the tick number.  Set the clock back (reset, set_tick) and an old stamp can match
    a later pair in which the env is NOT risky: it must then be honoured all the way (one-tick form in the pass,
    skipped by k_edge, moved by the restricted launch - which exits early only if NO tile was treated as risky)."""
    rng = np.random.RandomState(777)
    E = 12
    cfg = dict(m=3, n=3, length=200.0, capacity=20, rate=0.5)
    a = pairs_engine(E, **cfg)
    c = pertick_engine(E, **cfg)
    phase = rng.randint(2, size=(E, a.I)).astype(np.int32)

    def wild(envs, tick):
        """pathological cars in `envs` (risky for sure), nothing anywhere else; one decision of two ticks"""
        x, v, w, leading, lastcar = random_state(rng, E, a.R, 20, 200.0, crowd=0.9, beyond=1.6, sorted_x=False)
        keep = np.zeros(E, bool)
        keep[envs] = True
        x[~keep], v[~keep] = 0.0, 0.0
        leading[~keep], lastcar[~keep] = 1, 1
        for eng in (a, c):
            eng.reset(phase)
            eng.load_state(x, v, leading, lastcar)
            eng.set_tick(tick)
            eng.set_spawns(period=3)
            eng.set_actions(phase)
            eng.agent_step(2, remi=True)
        assert_engines_equal(a, c)
    wild(slice(0, 6), 40)                        # envs 0-5 stamped for the pair (40, 41)
    wild(slice(6, 12), 50)                       # envs 6-11 for (50, 51): "somebody is risky" now says 51
    xb, vb, lb, cb = wl.prefill_one_env(3, 3, 200.0, 20, 6, 12.0)
    for eng in (a, c):
        eng.reset(phase)
        eng.load_state(np.repeat(xb[None], E, 0), np.repeat(vb[None], E, 0), np.repeat(lb[None], E, 0), np.repeat(cb[None], E, 0))
        eng.set_tick(40)                         # nobody is risky now, but envs 0-5 still carry the stamp of tick 40
        eng.set_spawns(period=3)
        eng.set_actions(phase)
    ra = [t.clone() for t in a.agent_step(2, remi=True)]
    rc = [t.clone() for t in c.agent_step(2, remi=True)]
    for u, w_ in zip(ra, rc):
        assert torch.equal(u, w_)
    assert_engines_equal(a, c)
    assert a.tick == c.tick == 42


def test_first_split_call_of_new_handles():
    """The very first split call of a handle starts its second stream (a hardware queue of its own is created on that
    stream's first launch, which can take longer than a small batch's whole pair).  The second half must still begin
    at the tick the call began at: its clock is copied on the caller's stream ahead of the fork (copied on the second
    stream it raced with the first half's kernels, which move the clock on - seen once in a full run of this suite)."""
    if _TAIL[1] != "2":
        pytest.skip("split calls only")
    cfg = dict(m=3, n=3, length=120.0, capacity=14, rate=0.5)
    x, v, leading, lastcar = wl.prefill_one_env(3, 3, 120.0, 14, 8, 8.0)
    E = 6
    c = pertick_engine(E, **cfg)
    for trial in range(10):
        a = pairs_engine(E, **cfg)                   # a new handle, hence a new second stream, every time
        for eng in (a, c):
            eng.reset(np.zeros((E, eng.I), np.int32))
            eng.load_state(np.repeat(x[None], E, 0), np.repeat(v[None], E, 0), np.repeat(leading[None], E, 0),
                           np.repeat(lastcar[None], E, 0))
            eng.set_tick(17 + trial)
            eng.set_spawns(period=2)
            eng.set_actions(cycle_period=5)
        a.step(2)
        c.step(2)
        assert a.split_ticks() == 2 and a.tick == c.tick == 19 + trial
        assert_engines_equal(a, c)
