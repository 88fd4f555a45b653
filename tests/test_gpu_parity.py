"""GPU parity: the HIP path (through the C ABI, include/tfx.h) against the CPU oracle and the
golden vectors captured from the reference.

Bars (stated here, asserted below):
  * HIP vs oracle (oracle/idm_oracle.c): BIT-EXACT for everything - ring indices, obs, rewards,
    done, waiting, passed_dst AND the float32 x/v/w of every live car - free-running, any length.
    Both sides implement the same float contract (see the header of idm_oracle.c).
  * HIP vs golden (the reference itself): integers exact, x within 1 ulp and v within 2 ulp teacher-forced; integers
    exact for the first 120 ticks free-running (chaotic growth of the <=1-ulp power difference
    afterwards - see tests/test_oracle_golden.py).
"""
import os

import numpy as np
import pytest

from conftest import assert_floats_match_reference, golden_names, ulp_diff
from oracle.oracle import OracleEnv, live_mask

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


# every test of this module runs on both car layouts (TFX_TEST_LAYOUT=ring|transposed restricts)
LAYOUTS = [os.environ["TFX_TEST_LAYOUT"]] if os.environ.get("TFX_TEST_LAYOUT") else ["ring", "transposed"]
_LAYOUT = ["ring"]


@pytest.fixture(params=LAYOUTS, autouse=True)
def car_layout(request):
    _LAYOUT[0] = request.param
    yield request.param
    _LAYOUT[0] = "ring"


# ... and through both step paths: the LDS-resident multi-tick kernel k_res (what small envs get by
# default, packing 3 envs per workgroup so wavefronts straddle env boundaries) and the per-tick
# streaming kernels (TFX_RESIDENT=0: what big envs get)
@pytest.fixture(params=["resident", "resident1", "resident4", "resident_mixed", "pertick", "pairs"], autouse=True)
def step_path(request, monkeypatch):
    # "pairs": the per-tick kernels with two-tick passes (k_move_tt + k_edge) wherever a call has three ticks or more
    monkeypatch.setenv("TFX_PAIRS", "2" if request.param == "pairs" else "0")
    monkeypatch.setenv("TFX_TAIL", "2")        # (with the pairs: k_tail behind every pass, csrc/tfx_tail.hpp,
    monkeypatch.setenv("TFX_SPLIT", "2")       #  and the env range in two halves on two streams)
    if request.param.startswith("resident"):
        monkeypatch.setenv("TFX_RESIDENT", "1")
        monkeypatch.setenv("TFX_RES_EPB", "3")
        # lanes per road ("3": two, and four on the roads cars enter the map on - the handle's own choice for big batches)
        monkeypatch.setenv("TFX_RES_LPR", {"resident1": "1", "resident4": "4", "resident_mixed": "3"}.get(request.param, "2"))
    else:
        monkeypatch.setenv("TFX_RESIDENT", "0")
    yield request.param


def engine_for(g_or_cfg, n_envs=1, **kw):
    from gym_traffic.core import TfxEngine
    if isinstance(g_or_cfg, dict):
        c = g_or_cfg
    else:
        sc = g_or_cfg.sc
        c = dict(m=sc["m"], n=sc["n"], length=sc["L"], capacity=sc["C"], rate=sc["rate"],
                 learn_switch=sc["learn_switch"], validate=sc["mode"] == "validate",
                 entry_spec=0b1110 if sc["entry"] == "one" else 0)
    c = dict(c)
    c.update(kw)
    # ring: all three planes; transposed: the spawn-tick plane only where validate mode needs it
    planes = 3 if (_LAYOUT[0] == "ring" or c.get("validate")) else 2
    return TfxEngine(n_envs=n_envs, planes=planes, layout=_LAYOUT[0], **c)


def oracle_like(eng, **kw):
    return OracleEnv(eng.m, eng.n, float(eng.cfg.length), eng.C, eng.dest, eng.phases, eng.nexts,
                     n_envs=eng.E, rate=float(eng.cfg.rate), learn_switch=bool(eng.cfg.learn_switch),
                     validate=bool(eng.cfg.validate), **kw)


def counts(eng, road_lists):
    c = np.zeros((eng.E, max(1, eng.n_entry)), np.int32)
    for k, roads in enumerate(road_lists):
        for rd in roads:
            c[k, eng.entry_index[int(rd)]] += 1
    return c


def same_bits(a, b):
    """Bit-equal float32 arrays; NaNs compare equal to NaNs (payload/sign of a generated NaN is
    not part of the contract: x86 and gfx950 produce different default NaNs)."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return bool(np.all((a.view(np.int32) == b.view(np.int32)) | (np.isnan(a) & np.isnan(b))))


def assert_same_state(eng, orc, where=""):
    """Bit-exact comparison of the HIP engine with the oracle on everything that is defined."""
    ld, lc = eng.leading.cpu().numpy(), eng.lastcar.cpu().numpy()
    assert np.array_equal(ld, orc.leading), "leading " + where
    assert np.array_equal(lc, orc.lastcar), "lastcar " + where
    assert np.array_equal(eng.obs.cpu().numpy(), orc.obs), "obs " + where
    assert np.array_equal(eng.rewards.cpu().numpy(), orc.rewards), "rewards " + where
    assert np.array_equal(eng.waiting.cpu().numpy(), orc.waiting), "waiting " + where
    assert np.array_equal(eng.passed_dst.cpu().numpy(), orc.passed_dst), "passed_dst " + where
    st = eng.planes_numpy()
    rows = np.arange(eng.R)
    for k in range(eng.E):
        live = live_mask(ld[k], lc[k], eng.C)
        for plane, name, oplane in ((0, "x", orc.x[k]), (1, "v", orc.v[k]), (2, "w", orc.w[k])):
            if plane == 2 and eng.w is None:
                continue
            a, b = st[plane][k][live], oplane[live]
            assert same_bits(a, b), "%s env %d %s" % (name, k, where)
        # the fake leader's x sits in its slot, as in the reference
        a, b = st[0][k][rows, ld[k]], orc.x[k][rows, ld[k]]
        assert same_bits(a, b), "leader x env %d %s" % (k, where)


def assert_engines_equal(a, b):
    """Two HIP engines hold the same defined state: ring indices, counters and the (x, v, w) of
    every live car, bit for bit.  Dead ring slots hold junk by design and are not compared."""
    for name in ("leading", "lastcar", "obs", "rewards", "waiting", "passed_dst", "done"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    ld, lc = a.leading.cpu().numpy(), a.lastcar.cpu().numpy()
    pa, pb = a.planes_numpy(), b.planes_numpy()
    for k in range(a.E):
        live = live_mask(ld[k], lc[k], a.C)
        for u, v, name in zip(pa, pb, "xvw"):
            assert same_bits(u[k][live], v[k][live]), "%s env %d" % (name, k)
    assert (a.w is None) == (b.w is None)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", golden_names())
def test_free_running_vs_oracle_and_golden(name, golden_cache):
    """Every captured reference run, replayed through tfx_step: bit-equal to the oracle for the
    whole run, integer-equal to the reference for the first 120 ticks."""
    g = golden_cache(name)
    sc = g.sc
    eng = engine_for(g)
    orc = oracle_like(eng)
    assert np.array_equal(eng.nexts, g["nexts"]) and np.array_equal(eng.dest, g["dest"])
    assert np.array_equal(eng.phases, g["phases"]) and np.array_equal(eng.entrypoints, g["entrypoints"])
    eng.reset(g["init_phase"])
    orc.reset(g["init_phase"])
    ri = 0
    for t in range(sc["T"]):
        roads = g.spawns(t)
        eng.set_spawns(counts=counts(eng, [roads]))
        eng.set_actions(g["actions"][t][None, :])
        eng.step(1)
        _, _, odone = orc.step(g["actions"][t], [roads])
        k = t + 1
        assert np.array_equal(eng.done.cpu().numpy(), odone), (name, k)
        if k % 7 == 0 or k < 20 or k == sc["T"]:
            assert_same_state(eng, orc, "%s tick %d" % (name, k))
        if k <= 120:
            assert np.array_equal(eng.leading[0].cpu().numpy(), g["leading"][k]), (name, k)
            assert np.array_equal(eng.lastcar[0].cpu().numpy(), g["lastcar"][k]), (name, k)
            assert np.array_equal(eng.obs[0].cpu().numpy(), g["obs"][k]), (name, k)
            assert np.array_equal(eng.rewards[0].cpu().numpy(), g["rewards"][k]), (name, k)
            assert int(eng.done[0]) == int(g["done"][k]), (name, k)
        if sc["remi_every"] and k % sc["remi_every"] == 0:
            cor = eng.cars_on_roads()[0].cpu().numpy()
            assert np.array_equal(cor, orc.cars_on_roads()[0])
            rr = eng.remi_reward()[0].cpu().numpy()
            assert np.array_equal(rr, orc.remi_reward()[0])
            if k <= 120:
                assert np.array_equal(cor, g["cars_on_roads"][ri])
                assert np.array_equal(rr, g["remi_rewards"][ri])
            ri += 1
    if sc["mode"] == "validate":
        n = int(eng.n_trips[0])
        assert n == int(orc.n_trips[0]) and n > 10
        assert np.array_equal(eng.trip_times[0, :n].cpu().numpy(), orc.trip_times[0, :n])


@pytest.mark.parametrize("name", [n for n in golden_names() if "ints" not in n])
def test_teacher_forced_vs_golden(name, golden_cache):
    """Reference state at t -> one tfx_step -> reference state at t+1: ints exact, x <= 1 ulp, v <= 2 ulp
    (conftest.assert_floats_match_reference).
    Every fixture that carries car states, from every tick it carries (g4x4_cfg1 keeps the cars of
    every 10th tick only: there the integers of tick t+1 are checked, the floats are not)."""
    g = golden_cache(name)
    sc = g.sc
    eng = engine_for(g)
    at = {int(t): i for i, t in enumerate(g["state_ticks"])}
    for t in sorted(at):
        if t >= sc["T"]:
            continue
        i = at[t]
        eng.load_state(g["state_x"][i][None], g["state_v"][i][None], g["leading"][t][None],
                       g["lastcar"][t][None], w=g["state_w"][i][None])
        eng.obs.copy_(torch.as_tensor(g["obs"][t][None]))
        eng.rewards.copy_(torch.as_tensor(g["rewards"][t][None]))
        eng.waiting.copy_(torch.as_tensor(g["waiting"][t][None]))
        eng.passed_dst.copy_(torch.as_tensor(g["passed_dst"][t][None]))
        if sc["remi_every"] and t > 0 and t % sc["remi_every"] == 0:
            eng.remi_reward()
        eng.set_tick(t)
        eng.set_spawns(counts=counts(eng, [g.spawns(t)]))
        eng.set_actions(g["actions"][t][None, :])
        eng.step(1)
        k = t + 1
        ld, lc = eng.leading[0].cpu().numpy(), eng.lastcar[0].cpu().numpy()
        assert np.array_equal(ld, g["leading"][k]) and np.array_equal(lc, g["lastcar"][k]), (name, k)
        assert np.array_equal(eng.obs[0].cpu().numpy(), g["obs"][k]), (name, k)
        assert np.array_equal(eng.rewards[0].cpu().numpy(), g["rewards"][k]), (name, k)
        assert np.array_equal(eng.waiting[0].cpu().numpy(), g["waiting"][k]), (name, k)
        assert np.array_equal(eng.passed_dst[0].cpu().numpy(), g["passed_dst"][k]), (name, k)
        assert int(eng.done[0]) == int(g["done"][k])
        if k not in at:
            continue
        sx, sv, sw = [a[0] for a in eng.planes_numpy()]
        live = live_mask(ld, lc, sc["C"])
        if live.any():
            assert_floats_match_reference(sx[live], sv[live], g["state_x"][at[k]][live], g["state_v"][at[k]][live],
                                          rate=sc["rate"], where=(name, k), single_default_archetype=True)
            if eng.w is not None:
                assert np.array_equal(sw[live], g["state_w"][at[k]][live])


# ------------------------------------------------------------------------------------------------
def random_state(rng, E, R, C, length, crowd=0.5, beyond=0.15, sorted_x=True):
    """Random ring states incl. wrapped rings, empty and full roads, cars past the road end."""
    x = np.zeros((E, R, C), np.float32)
    v = np.zeros((E, R, C), np.float32)
    w = np.zeros((E, R, C), np.float32)
    leading = rng.randint(1, C, size=(E, R)).astype(np.int32)
    n = np.minimum(rng.binomial(C - 2, crowd, size=(E, R)), C - 2)
    n[rng.rand(E, R) < 0.1] = 0
    n[rng.rand(E, R) < 0.1] = C - 2
    lastcar = leading.copy()
    for k in range(E):
        for e in range(R):
            cnt = int(n[k, e])
            pos = np.sort(rng.uniform(-20, length * (1 + beyond), size=cnt))[::-1] if sorted_x \
                else rng.uniform(-20, length * (1 + beyond), size=cnt)
            s = int(leading[k, e])
            for j in range(cnt):
                s = s + 1 if s + 1 < C else 1
                x[k, e, s] = pos[j]
                v[k, e, s] = rng.choice([0.0, rng.uniform(0, 15)])
                w[k, e, s] = rng.randint(0, 50)
            lastcar[k, e] = s
            # exit roads keep the +inf leader they got at reset (traffic_env.py:263; update_lights
            # never touches them); train roads get theirs rewritten every tick
            x[k, e, leading[k, e]] = np.inf
    return x, v, w, leading, lastcar


def load_both(eng, orc, x, v, w, leading, lastcar, phase, elapsed):
    eng.load_state(x, v, leading, lastcar, w=w)
    r, I = eng.r, eng.I
    obs = np.zeros((eng.E, eng.obs_len), np.int32)
    obs[:, 2 * r:2 * r + I] = phase
    obs[:, 2 * r + I:] = elapsed
    eng.obs.copy_(torch.as_tensor(obs))
    eng.waiting.zero_()
    eng.passed_dst.zero_()
    eng.rewards.zero_()
    for k in range(eng.E):
        orc.load_planes(k, x[k], v[k], w[k], leading[k], lastcar[k])
    orc.obs[:] = obs
    orc.waiting[:] = 0
    orc.passed_dst[:] = 0
    orc.rewards[:] = 0


@pytest.mark.parametrize("m,n,C,length,validate", [(2, 2, 10, 60.0, False), (3, 2, 20, 120.0, True),
                                                   (4, 4, 34, 200.0, False), (2, 3, 66, 400.0, False),
                                                   (2, 2, 130, 800.0, True), (2, 2, 258, 900.0, False)])
@pytest.mark.parametrize("sorted_x", [True, False])
def test_random_states_vs_oracle(m, n, C, length, validate, sorted_x):
    """Arbitrary ring states (wrapped, full, empty, many cars beyond the road end so that several
    pop at once and rings overflow): a few free-running ticks must stay bit-equal to the oracle.
    Exercises the serial-advance fallback and the 1/2/4-wave-per-road kernels."""
    rng = np.random.RandomState(1234 + C + int(sorted_x))
    E = 6
    eng = engine_for(dict(m=m, n=n, length=length, capacity=C, rate=0.5, validate=validate), n_envs=E)
    orc = oracle_like(eng)
    for trial in range(4):
        # beyond = 1.6: some cars sit more than a road length past the end, so a handed-off car is
        # popped again downstream in the same tick (the serial-advance path of either layout)
        x, v, w, leading, lastcar = random_state(rng, E, eng.R, C, length, crowd=rng.choice([0.3, 0.8]),
                                                 beyond=rng.choice([0.0, 0.05, 0.4, 1.6]), sorted_x=sorted_x)
        phase = rng.randint(2, size=(E, eng.I)).astype(np.int32)
        elapsed = rng.randint(0, 12, size=(E, eng.I)).astype(np.int32)
        load_both(eng, orc, x, v, w, leading, lastcar, phase, elapsed)
        eng.set_tick(60)
        orc.steps[:] = 60
        orc.n_trips[:] = 0
        if validate:
            eng.n_trips.zero_()
        for t in range(6):
            act = rng.randint(2, size=(E, eng.I)).astype(np.int32)
            roads = [rng.choice(eng.entrypoints, size=rng.randint(0, 4)).tolist() for _ in range(E)]
            eng.set_spawns(counts=counts(eng, roads))
            eng.set_actions(act)
            eng.step(1)
            _, _, odone = orc.step(act, roads)
            assert np.array_equal(eng.done.cpu().numpy(), odone), (trial, t)
            assert_same_state(eng, orc, "trial %d tick %d" % (trial, t))
        if validate:
            nt = eng.n_trips.cpu().numpy()
            assert np.array_equal(nt, orc.n_trips)
            for k in range(E):
                assert np.array_equal(eng.trip_times[k, :nt[k]].cpu().numpy(), orc.trip_times[k, :nt[k]])


@pytest.mark.parametrize("variant", ["91"])
def test_streaming_move_kernels_on_random_states(variant, monkeypatch, car_layout, step_path):
    """The one-wavefront-per-tile kernel (k_move_t, the default for launches that fill the chip) forced at test
    sizes, where the launch heuristics would pick the four-waves-per-tile kernel: pathological ring states (cars past
    the end, unsorted, NaN-producing zero gaps, huge speeds that leave the fast domain) and ordinary
    traffic, bit-equal to the oracle."""
    if car_layout != "transposed" or step_path != "pertick":
        pytest.skip("transposed-layout per-tick kernels")
    monkeypatch.setenv("TFX_MOVE_VARIANT", variant)
    rng = np.random.RandomState(77 + int(variant))
    for (m, n, C, length, E) in [(2, 2, 10, 60.0, 5), (3, 3, 34, 200.0, 40), (2, 3, 66, 400.0, 9), (2, 2, 130, 300.0, 3)]:
        eng = engine_for(dict(m=m, n=n, length=length, capacity=C, rate=0.5), n_envs=E)
        orc = oracle_like(eng)
        for trial in range(3):
            x, v, w, leading, lastcar = random_state(rng, E, eng.R, C, length, crowd=rng.choice([0.3, 0.8]),
                                                     beyond=rng.choice([0.0, 0.05, 0.4, 1.6]), sorted_x=bool(trial % 2))
            if trial == 2:
                # leave the fast domain on purpose: enormous and denormal speeds, exact-zero gap denominators
                v[rng.rand(*v.shape) < 0.02] = 3e7
                v[rng.rand(*v.shape) < 0.02] = 1e-30
                pick = rng.rand(*x[:, :, 2:].shape) < 0.05          # follower exactly one car length behind:
                x[:, :, 2:][pick] = (x[:, :, 1:-1] - np.float32(4.0))[pick]   # gap denominator = eps
                np.put_along_axis(x, leading[:, :, None].astype(np.int64), np.inf, axis=2)   # (fake leaders stay at +inf)
            phase = rng.randint(2, size=(E, eng.I)).astype(np.int32)
            elapsed = rng.randint(0, 12, size=(E, eng.I)).astype(np.int32)
            load_both(eng, orc, x, v, w, leading, lastcar, phase, elapsed)
            eng.set_tick(60)
            orc.steps[:] = 60
            for t in range(8):
                act = rng.randint(2, size=(E, eng.I)).astype(np.int32)
                roads = [rng.choice(eng.entrypoints, size=rng.randint(0, 4)).tolist() for _ in range(E)]
                eng.set_spawns(counts=counts(eng, roads))
                eng.set_actions(act)
                eng.step(1)
                _, _, odone = orc.step(act, roads)
                assert np.array_equal(eng.done.cpu().numpy(), odone), (variant, trial, t)
                assert_same_state(eng, orc, "variant %s C=%d trial %d tick %d" % (variant, C, trial, t))


def test_kernel_halves_vs_oracle():
    """tfx_move_cars and tfx_advance_finished_cars on their own (the numba-signature level of the
    boundary, traffic_env.py:187-191 and :117-120)."""
    rng = np.random.RandomState(7)
    E, C, length = 4, 20, 150.0
    eng = engine_for(dict(m=3, n=3, length=length, capacity=C, rate=0.5), n_envs=E)
    orc = oracle_like(eng)
    x, v, w, leading, lastcar = random_state(rng, E, eng.R, C, length, crowd=0.6, beyond=0.1)
    phase = rng.randint(2, size=(E, eng.I)).astype(np.int32)
    elapsed = rng.randint(0, 12, size=(E, eng.I)).astype(np.int32)
    load_both(eng, orc, x, v, w, leading, lastcar, phase, elapsed)
    eng.set_spawns()
    eng.set_actions(phase)            # action == phase: no change, elapsed + 1
    eng.move_cars()
    orc.elapsed[:] += 1
    orc.move_cars()
    if eng.layout == "ring":
        # (the transposed kernel already compacts the roads while it moves the cars, so between
        # the two halves its cars are not addressable by the not-yet-advanced ring indices)
        sx, sv, _ = eng.planes_numpy()
        for k in range(E):
            live = live_mask(leading[k], lastcar[k], C)
            assert same_bits(sx[k][live], orc.x[k][live])
            assert same_bits(sv[k][live], orc.v[k][live])
    assert np.array_equal(eng.waiting.cpu().numpy(), orc.waiting)
    assert np.array_equal(eng.detected.cpu().numpy(), orc.detected)
    eng.advance_finished_cars()
    orc.passed[:] = 0
    orc.rewards[:] = 0
    odone = orc.advance()
    assert np.array_equal(eng.done.cpu().numpy(), odone)
    assert_same_state(eng, orc, "after advance")


def test_batched_poisson_rollout_vs_oracle():
    """cfg1 shape (4x4, 32 cars/road): 16 envs with their own seeded Poisson arrivals and random
    light actions, 250 ticks, bit-equal to the oracle throughout."""
    from gym_traffic.spawner import SpawnSchedule
    E, T = 16, 250
    eng = engine_for(dict(m=4, n=4, length=200.0, capacity=34, rate=0.5), n_envs=E)
    orc = oracle_like(eng)
    rng = np.random.RandomState(99)
    cps = 0.3 * 4 * 4
    sched = [SpawnSchedule(np.random.RandomState(100 + k), k % 2 == 0, eng.entrypoints, lambda: (cps, 0.5))
             for k in range(E)]
    ph = rng.randint(2, size=(E, eng.I)).astype(np.int32)
    eng.reset(ph)
    orc.reset(ph)
    act = None
    overflow_ticks = 0
    for t in range(T):
        if t % 10 == 0:
            act = rng.randint(2, size=(E, eng.I)).astype(np.int32)
        roads = [s.next_tick() for s in sched]
        eng.set_spawns(counts=counts(eng, roads))
        eng.set_actions(act)
        eng.step(1)
        _, _, odone = orc.step(act, roads, nthreads=4)
        assert np.array_equal(eng.done.cpu().numpy(), odone), t
        overflow_ticks += int(odone.sum())
        if t % 10 == 9:
            assert_same_state(eng, orc, "tick %d" % t)
            assert np.array_equal(eng.remi_reward().cpu().numpy(), orc.remi_reward())
    assert_same_state(eng, orc, "end")
    assert overflow_ticks > 0          # overflow + penalty path was exercised


def test_multi_tick_call_equals_single_ticks():
    """tfx_step(n) == n x tfx_step(1) with per-tick spawn/action buffers."""
    rng = np.random.RandomState(5)
    E, T = 3, 12
    cfg = dict(m=2, n=3, length=100.0, capacity=12, rate=0.5)
    a = engine_for(cfg, n_envs=E)
    b = engine_for(cfg, n_envs=E)
    ph = rng.randint(2, size=(E, a.I)).astype(np.int32)
    acts = rng.randint(2, size=(T, E, a.I)).astype(np.int32)
    sp = rng.randint(0, 2, size=(T, E, a.n_entry)).astype(np.int32)
    a.reset(ph)
    b.reset(ph)
    a.set_actions(acts, per_tick=True)
    a.set_spawns(counts=sp, per_tick=True)
    a.step(T)
    for t in range(T):
        b.set_actions(acts[t])
        b.set_spawns(counts=sp[t])
        b.step(1)
    assert_engines_equal(a, b)
    assert a.tick == b.tick == T


def test_on_device_controllers_match_host_rule():
    """TFX_SPAWN_PERIODIC / TFX_ACTION_CYCLE (bench inputs) == the same rule fed from the host."""
    E, T, period_s, period_a = 5, 50, 8, 20
    cfg = dict(m=3, n=3, length=150.0, capacity=18, rate=0.5)
    a = engine_for(cfg, n_envs=E)
    b = engine_for(cfg, n_envs=E)
    ph = np.zeros((E, a.I), np.int32)
    a.reset(ph)
    b.reset(ph)
    a.set_spawns(period=period_s)
    a.set_actions(cycle_period=period_a)
    a.step(T)
    for t in range(T):
        c = np.zeros((E, a.n_entry), np.int32)
        for j, rd in enumerate(a.entrypoints):
            if t % period_s == rd % period_s:
                c[:, j] = 1
        act = np.stack([np.full(a.I, ((t + k % period_a) // period_a) & 1, np.int32) for k in range(E)])
        b.set_spawns(counts=c)
        b.set_actions(act)
        b.step(1)
    assert_engines_equal(a, b)
    assert int(a.cars_on_roads_flat().sum()) > 50


def test_vehicle_update_counter():
    """tfx_vehicle_updates == sum over ticks of the live cars move_cars advanced
    (cars present before the tick + cars spawned in it; no ring overflows in this run)."""
    E, period = 4, 4
    eng = engine_for(dict(m=2, n=2, length=100.0, capacity=30, rate=0.5), n_envs=E)
    eng.reset(np.zeros((E, eng.I), np.int32))
    eng.set_spawns(period=period)
    eng.set_actions(cycle_period=10)
    eng.reset_counters()
    expect = 0
    for t in range(30):
        before = int(eng.cars_on_roads_flat().sum())
        spawned = E * sum(1 for rd in eng.entrypoints if t % period == rd % period)
        eng.step(1)
        assert int(eng.done.sum()) == 0
        expect += before + spawned
    assert eng.vehicle_updates() == expect
    assert expect > 500


def test_reciprocal_division_selftest():
    """The move kernel replaces two IEEE divisions by constant divisors with a reciprocal form only
    after checking on the device, for every admitted numerator, that the quotient is bit-identical."""
    eng = engine_for(dict(m=2, n=2, length=100.0, capacity=10, rate=0.5))
    st = eng.fastdiv_status()
    assert st["mismatches"] == 0 and st["enabled"]


@pytest.mark.parametrize("m,n,C", [(1, 1, 3), (1, 3, 3), (3, 1, 4), (1, 1, 258), (2, 1, 5)])
def test_minimal_and_maximal_shapes_vs_oracle(m, n, C):
    """The corners of the parameter space: one-intersection and one-row grids, rings that hold a
    single car (CAPACITY 3 = scratch slot + fake leader + one car), the largest ring (256 cars)."""
    rng = np.random.RandomState(77 + 7 * m + n + C)
    E, length = 3, 45.0
    eng = engine_for(dict(m=m, n=n, length=length, capacity=C, rate=0.5), n_envs=E)
    orc = oracle_like(eng)
    ph = rng.randint(2, size=(E, eng.I)).astype(np.int32)
    eng.reset(ph)
    orc.reset(ph)
    dones = 0
    for t in range(90):
        act = rng.randint(2, size=(E, eng.I)).astype(np.int32) if t % 5 == 0 else act
        roads = [rng.choice(eng.entrypoints, size=rng.randint(0, 3)).tolist() for _ in range(E)]
        eng.set_spawns(counts=counts(eng, roads))
        eng.set_actions(act)
        eng.step(1)
        _, _, odone = orc.step(act, roads)
        assert np.array_equal(eng.done.cpu().numpy(), odone), t
        dones += int(odone.sum())
        if t % 6 == 0 or t == 89:
            assert_same_state(eng, orc, "tick %d" % t)
    if C <= 5:
        assert dones > 0        # single-car rings overflow all the time: the penalty path ran
