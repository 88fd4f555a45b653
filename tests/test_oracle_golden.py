"""Pins the CPU oracle (oracle/idm_oracle.c) against golden vectors captured from the
reference's own TrafficEnv (oracle/gen_golden.py; /root/reference/gym_traffic/envs/traffic_env.py).

Two tiers:
  * teacher-forced: load the reference's state at tick t, step once with the same action and
    spawns, compare with the reference's tick t+1.  Every integer output must be EXACT, x within
    1 ulp, v within 2 ulp or a rate 2^-22 absolute (conftest.assert_floats_match_reference: the
    reference's NumPy float32 `**` is a platform SIMD routine within 1 ulp of the correctly rounded
    power the oracle's contract uses; one ulp of the power term can flip three later roundings).
  * free-running from reset: integers exact for the first FREE_TICKS ticks.  The model is
    chaotic near standstill (a 1-ulp difference grows ~4x per tick once a car brakes hard behind
    its leader), so bit-equal integers cannot hold for an unbounded horizon; every fixture stays
    exact for at least 130 ticks, we assert 120.
"""
import numpy as np
import pytest

from conftest import assert_floats_match_reference, golden_names, ulp_diff
from oracle.oracle import OracleEnv, live_mask

TF_ULP = 1
FREE_TICKS = 120


def make_env(g, n_envs=1):
    sc = g.sc
    return OracleEnv(sc["m"], sc["n"], sc["L"], sc["C"], g["dest"], g["phases"], g["nexts"],
                     n_envs=n_envs, rate=sc["rate"], learn_switch=sc["learn_switch"],
                     validate=sc["mode"] == "validate")


def step(env, g, t):
    """One oracle tick with the fixture's inputs of tick t (incl. the archetype row of every spawned car)."""
    sa = g.spawn_archs(t)
    return env.step(g["actions"][t], [g.spawns(t)], spawn_arch=None if sa is None else [sa], archetypes=g.archetypes)


def ints_equal(env, obs, rew, done, g, k):
    return [n for n, a, b in (
        ("leading", env.leading[0], g["leading"][k]), ("lastcar", env.lastcar[0], g["lastcar"][k]),
        ("obs", obs[0], g["obs"][k]), ("rewards", rew[0], g["rewards"][k]),
        ("done", int(done[0]), int(g["done"][k])), ("waiting", env.waiting[0], g["waiting"][k]),
        ("passed_dst", env.passed_dst[0], g["passed_dst"][k])) if not np.array_equal(a, b)]


@pytest.mark.parametrize("name", golden_names(None))
def test_free_running_integers(name, golden_cache):
    g = golden_cache(name)
    env = make_env(g)
    env.reset(g["init_phase"])
    ri = 0
    for t in range(min(FREE_TICKS, g.sc["T"])):
        obs, rew, done = step(env, g, t)
        k = t + 1
        assert ints_equal(env, obs, rew, done, g, k) == [], (name, k)
        if g.sc["remi_every"] and k % g.sc["remi_every"] == 0:
            assert np.array_equal(env.cars_on_roads()[0], g["cars_on_roads"][ri])
            assert np.array_equal(env.remi_reward()[0], g["remi_rewards"][ri])
            ri += 1
    assert ri > 0


@pytest.mark.parametrize("name", [n for n in golden_names(None) if "ints" not in n])
def test_teacher_forced_every_tick(name, golden_cache):
    """From every tick whose car states the fixture carries (all of them for the 2x2 / 3x3 runs, every 10th for
    g4x4_cfg1, ticks 20 j and 20 j + 1 for g8x8_c130): integers of tick t+1 exact, floats within TF_ULP wherever the
    fixture also carries tick t+1."""
    g = golden_cache(name)
    sc = g.sc
    env = make_env(g)
    rows = np.arange(env.R)
    wrapped_seen = floats_checked = 0
    at = {int(t): i for i, t in enumerate(g["state_ticks"])}
    for t in sorted(at):
        if t >= sc["T"]:
            continue
        i = at[t]
        env.load_planes(0, g["state_x"][i], g["state_v"][i], g["state_w"][i], g["leading"][t], g["lastcar"][t],
                        arch=g["state_a"][i] if g.archetypes is not None else None, archetypes=g.archetypes)
        env.obs[0] = g["obs"][t]
        env.rewards[0] = g["rewards"][t]
        env.waiting[0] = g["waiting"][t]
        env.passed_dst[0] = g["passed_dst"][t]
        if sc["remi_every"] and t > 0 and t % sc["remi_every"] == 0:
            env.remi_reward()          # the capture called remi_reward() after recording tick t
        env.steps[0] = t
        wrapped_seen += int((g["leading"][t] > g["lastcar"][t]).sum())
        obs, rew, done = step(env, g, t)
        k = t + 1
        assert ints_equal(env, obs, rew, done, g, k) == [], (name, k)
        assert np.array_equal(env.x[0][rows, env.leading[0]], g["leader_x"][k])
        if k not in at:
            continue
        x, v, w = env.planes(0)
        live = live_mask(env.leading[0], env.lastcar[0], sc["C"])
        if live.any():
            a_max = 3.0 if g.archetypes is None else float(g.archetypes[:, 3].max())
            assert_floats_match_reference(x[live], v[live], g["state_x"][at[k]][live], g["state_v"][at[k]][live],
                                          a_max=a_max, rate=sc["rate"], where=(name, k),
                                          single_default_archetype=g.archetypes is None)
            assert np.array_equal(w[live], g["state_w"][at[k]][live])
            if g.archetypes is not None:   # every car still carries the row it was spawned from, through every handoff
                assert np.array_equal(env.arch_plane(0, g.archetypes)[live], g["state_a"][at[k]][live])
            floats_checked += int(live.sum())
    if name == "g3x3_default":
        assert wrapped_seen > 100      # the wrapped-ring branch (traffic_env.py:202-212) is exercised
    if name == "g8x8_c130":            # rings longer than one wavefront, wrapped, compared car by car
        assert wrapped_seen > 300 and floats_checked > 50000


@pytest.mark.parametrize("name", ["g2x2_s0_poi_c10", "g2x2_s0_reg_c20"])
def test_move_cars_alone_against_mid_state(name, golden_cache):
    """Kernel-level: move_cars (traffic_env.py:187-212) without the advance."""
    g = golden_cache(name)
    sc = g.sc
    env = make_env(g)
    for t in range(0, sc["T"], 3):
        env.load_planes(0, g["state_x"][t], g["state_v"][t], g["state_w"][t], g["leading"][t], g["lastcar"][t])
        env.obs[0] = g["obs"][t]
        env.waiting[0] = g["waiting"][t]
        env.passed_dst[0] = g["passed_dst"][t]
        if t > 0 and t % sc["remi_every"] == 0:
            env.remi_reward()
        env.steps[0] = t
        # replay the phase update + spawns, then move_cars only: do it through step() on a copy of
        # the state with an advance that cannot fire is not possible, so compare via the full
        # step's pre-advance snapshot instead: run move only when no spawn happened this tick.
        if len(g.spawns(t)):
            continue
        cur, el = env.current_phase[0], env.elapsed[0]
        change = (cur != 0) != (g["actions"][t] != 0)
        cur[:] = g["actions"][t]
        el[:] = (el + 1) * (~change)
        env.move_cars()
        live = live_mask(env.leading[0], env.lastcar[0], sc["C"])
        if live.any():
            assert ulp_diff(env.x[0][live], g["mid_x"][t][live]).max() <= TF_ULP
            assert ulp_diff(env.v[0][live], g["mid_v"][t][live]).max() <= TF_ULP


def test_trip_times_validate_mode(golden_cache):
    """advance_hack (traffic_env.py:139-157): (tick - w) / 2 for cars leaving the map."""
    g = golden_cache("g2x2_validate")
    env = make_env(g)
    env.reset(g["init_phase"])
    for t in range(200):
        env.step(g["actions"][t], [g.spawns(t)])
        assert int(env.n_trips[0]) == int(g["trip_count"][t + 1])
        if (t + 1) % 10 == 0:
            env.remi_reward()
    n = int(env.n_trips[0])
    assert n > 10
    assert np.array_equal(env.trip_times[0, :n].astype(np.float64), g["trip_times"][:n])


def test_known_answer_anchor(golden_cache):
    """SURVEY.md Appendix A anchor: GridRoad(2,2,250) static tables."""
    g = golden_cache("g2x2_s0_poi_c20")
    assert g["nexts"].tolist() == [1, 18, 3, 19, 22, 4, 23, 6, 10, 11, 20, 21, 16, 17, 12, 13] + [-1] * 8
    assert g["dest"].tolist() == [0, 1, 2, 3] * 4 + [-1] * 8
    assert g["phases"].tolist() == [1] * 8 + [0] * 16
    assert g["entrypoints"].tolist() == [0, 2, 5, 7, 8, 9, 14, 15]


def test_batched_envs_are_independent(golden_cache):
    """E > 1 in the oracle = E copies of the reference env: env k must equal a solo run."""
    g = golden_cache("g2x2_s1_poi_c10")
    solo = make_env(g)
    solo.reset(g["init_phase"])
    bat = make_env(g, n_envs=3)
    bat.reset(g["init_phase"])
    for t in range(60):
        solo.step(g["actions"][t], [g.spawns(t)])
        bat.step(g["actions"][t], [g.spawns(t), [], g.spawns(t)], nthreads=2)
    for k in (0, 2):
        assert np.array_equal(bat.state[k], solo.state[0])
        assert np.array_equal(bat.leading[k], solo.leading[0])
        assert np.array_equal(bat.obs[k], solo.obs[0])
    assert not np.array_equal(bat.leading[1], solo.leading[0])
