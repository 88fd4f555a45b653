"""NumPy-batched restatement of the reference's env tick - the second CPU baseline of SURVEY.md 8(d).

TEST INFRASTRUCTURE / BASELINE ONLY (like everything under oracle/): bench.py's cpu_baseline leg times it on ONE
core beside the C port, tests/test_numpy_env.py checks it bit for bit against oracle/idm_oracle.c.  The product
package never imports it.

It is what a NumPy user would write to batch the reference over E envs without leaving Python: the arrays of
gym_traffic/envs/traffic_env.py with a leading env dimension, `sim` (:50-62) as whole-array arithmetic over
every car of every road of every env at once (cars addressed in ring order through index arrays), `update_lights`
(:81-94) over [E, r], and the two order-dependent parts - `add_car` (:97-114) and `advance_finished_cars`
(:117-135) - vectorised over envs but walked road by road, in the reference's order.  Single archetype
(traffic_env.py:35-43); the float contract is the oracle's (binary32, q**4 rounded once from binary64).
"""
import time

import numpy as np

F = np.float32


class NumpyBatchedEnv(object):
    def __init__(self, m, n, length, capacity, dest, phases, nexts, n_envs=1, rate=0.5,
                 v=11.11, l=4.0, a=3.0, v0=13.89, b=6.0, T=2.0, s0=1.0):
        self.m, self.n, self.E, self.C = int(m), int(n), int(n_envs), int(capacity)
        self.I = self.m * self.n
        self.r = 4 * self.I
        self.R = self.r + 2 * self.m + 2 * self.n
        self.dest, self.phases, self.nexts = (np.asarray(t, np.int64) for t in (dest, phases, nexts))
        self.length, self.rate = F(length), F(rate)
        self.cv, self.cl, self.ca, self.cv0, self.cb, self.cT, self.cs0 = (F(t) for t in (v, l, a, v0, b, T, s0))
        self.two_sab = F(2.0) * np.sqrt(self.ca * self.cb, dtype=F)
        self.eps, self.thresh, self.near_end = F(1e-8), F(0.2), F(self.length - F(10.0))
        self.yellow, self.ovf_pen = 6, F(10.0)
        E, R, C, I, r = self.E, self.R, self.C, self.I, self.r
        self.x = np.zeros((E, R, C), F)
        self.v = np.zeros((E, R, C), F)
        self.leading = np.ones((E, R), np.int64)
        self.lastcar = np.ones((E, R), np.int64)
        self.obs = np.zeros((E, 2 * r + 2 * I), np.int32)
        self.rewards = np.zeros((E, I), F)
        self.waiting = np.zeros((E, r), np.int32)
        self.passed_dst = np.zeros((E, I), bool)
        self.done = np.zeros(E, bool)
        self.vehicle_updates = 0
        self._k = np.arange(C - 1, dtype=np.int64)[None, None, :]          # position behind the fake leader
        self._env = np.arange(E)

    # views with the reference's names (traffic_env.py:372-376)
    @property
    def passed(self):
        return self.obs[:, :self.r]

    @property
    def detected(self):
        return self.obs[:, self.r:2 * self.r]

    @property
    def current_phase(self):
        return self.obs[:, 2 * self.r:2 * self.r + self.I]

    @property
    def elapsed(self):
        return self.obs[:, 2 * self.r + self.I:]

    def reset(self, phase_init):
        self.x[:, :, 1] = np.inf
        self.v[:, :, 1] = 0
        self.leading[:] = 1
        self.lastcar[:] = 1
        self.obs[:] = 0
        self.current_phase[:] = np.asarray(phase_init, np.int32)
        self.waiting[:] = 0
        self.passed_dst[:] = False

    def counts(self):
        ld, lc = self.leading, self.lastcar
        return lc - ld + (ld > lc) * (self.C - 1)

    # ---- traffic_env.py:97-114 for one road index per selected env ---------------------------------
    def _add_car(self, envs, road, x_car, v_car):
        """envs: int array; road: int array (same length); pushes (x_car, v_car); returns overflow mask."""
        ld, lc = self.leading[envs, road], self.lastcar[envs, road]
        pos = np.where(lc + 1 >= self.C, 1, lc + 1)
        start = np.where(lc != ld, (self.x[envs, road, lc] - self.cl) - self.cs0, F(np.inf)).astype(F)
        ok = pos != ld
        e_ok, r_ok, p_ok = envs[ok], road[ok], pos[ok]
        self.x[e_ok, r_ok, p_ok] = np.minimum(np.broadcast_to(x_car, ld.shape)[ok], start[ok])
        self.v[e_ok, r_ok, p_ok] = np.broadcast_to(v_car, ld.shape)[ok]
        self.lastcar[e_ok, r_ok] = p_ok
        bad = ~ok
        dst = self.dest[road]
        pen = bad & (dst >= 0)
        np.subtract.at(self.rewards, (envs[pen], dst[pen]), self.ovf_pen)
        return bad

    def step(self, action, spawn_counts=None, entrypoints=None):
        """action int [E, I]; spawn_counts int [E, n_entry] cars per entry road this tick (entrypoints lists the
        roads).  One TrafficEnv._step (traffic_env.py:224-248) for every env."""
        E, R, C, I, r = self.E, self.R, self.C, self.I, self.r
        act = np.broadcast_to(np.asarray(action, np.int32), (E, I))
        cur, el = self.current_phase, self.elapsed
        change = (cur != 0) != (act != 0)
        cur[:] = act
        el[:] = (el + 1) * (~change)
        self.rewards[:] = 0
        self.passed[:] = 0
        overflowed = np.zeros(E, bool)
        if spawn_counts is not None:
            sc = np.asarray(spawn_counts)
            for j in range(int(sc.max()) if sc.size else 0):
                ee, jj = np.nonzero(sc > j)
                bad = self._add_car(ee, np.asarray(entrypoints, np.int64)[jj], F(0.0), self.cv)
                np.logical_or.at(overflowed, ee, bad)
        self._move_cars()
        overflowed |= self._advance()
        self.done[:] = overflowed
        return self.obs, self.rewards, self.done

    # ---- traffic_env.py:187-212 (update_lights :81-94, sim :50-62) for every road of every env -------
    def _move_cars(self):
        E, R, C, I, r = self.E, self.R, self.C, self.I, self.r
        ld, lc = self.leading, self.lastcar
        env = self._env[:, None]
        tr = np.arange(r)[None, :]
        dst = self.dest[:r][None, :]
        red = (self.phases[:r][None, :] == self.current_phase[env, dst]) | (self.elapsed[env, dst] < self.yellow)
        nr = self.nexts[:r][None, :]
        nonempty = lc[env, nr] != ld[env, nr]
        tail = self.x[env, nr, lc[env, nr]] + self.length
        self.x[env, tr, ld[:, :r]] = np.where(red, self.length, np.where(nonempty, tail, F(np.inf))).astype(F)

        n = self.counts()
        self.vehicle_updates += int(n.sum())
        k = self._k
        live = k < n[:, :, None]
        slot = ld[:, :, None] + 1 + k
        slot = np.where(slot >= C, slot - (C - 1), slot)
        slot = np.where(live, slot, 0)                       # dead positions read slot 0 (scratch), never written
        lslot = np.where(k == 0, ld[:, :, None], np.roll(slot, 1, axis=2))
        x = np.take_along_axis(self.x, slot, 2)
        v = np.take_along_axis(self.v, slot, 2)
        xl = np.take_along_axis(self.x, lslot, 2)
        vl = np.where(k == 0, F(0.0), np.take_along_axis(self.v, lslot, 2)).astype(F)
        ll = np.where(k == 0, F(0.0), self.cl).astype(F)
        with np.errstate(all="ignore"):
            t = v * self.cT + (v * (v - vl)) / self.two_sab
            s_star = self.cs0 + np.where(F(0.0) >= t, F(0.0), t)
            s = (xl - x) - ll
            q = (v / self.cv0).astype(np.float64)
            q2 = q * q
            qd = (q2 * q2).astype(F)
            u = s_star / (s + self.eps)
            dv = self.ca * ((F(1.0) - qd) - u * u)
            dvr = dv * self.rate
            dx = self.rate * v + (F(0.5) * dvr) * self.rate
            xn = x + np.where(dx > 0, dx, F(0.0) * dx)
            t2 = v + dvr
            vn = np.where(F(0.0) >= t2, F(0.0), t2)
        ee, rr, kk = np.nonzero(live)
        ss = slot[ee, rr, kk]
        self.x[ee, rr, ss] = xn[ee, rr, kk]
        self.v[ee, rr, ss] = vn[ee, rr, kk]
        # waiting / detected (:199-201, :208-212; the wrapped ring's second segment tests x, not v)
        second = (ld > lc)[:, :, None] & (slot <= lc[:, :, None])
        wq = np.where(second, xn, vn)
        w = (live & (wq < self.thresh))[:, :r].sum(2)
        dcount = (live & (xn > self.near_end))[:, :r].sum(2)
        has = n[:, :r] > 0
        self.waiting += np.where(has, w, 0).astype(np.int32)
        self.detected[:] = np.where(has, dcount, self.detected)

    # ---- traffic_env.py:117-135, road by road in the reference's order, vectorised over envs -------
    def _advance(self):
        C = self.C
        overflowed = np.zeros(self.E, bool)
        for e in range(self.R):
            nr = int(self.nexts[e])
            while True:
                ld, lc = self.leading[:, e], self.lastcar[:, e]
                head = np.where(ld + 1 >= C, 1, ld + 1)
                go = (ld != lc) & (self.x[self._env, e, head] > self.length)
                if not go.any():
                    break
                ee = self._env[go]
                hh = head[go]
                if nr >= 0:
                    self.passed[ee, e] += 1
                    self.passed_dst[ee, self.dest[e]] = True
                    xc = self.x[ee, e, hh] - self.length
                    bad = self._add_car(ee, np.full(ee.shape, nr, np.int64), xc, self.v[ee, e, hh])
                    overflowed[ee] |= bad
                self.x[ee, e, hh] = self.x[ee, e, ld[go]]
                self.v[ee, e, hh] = 0
                self.leading[ee, e] = hh
        return overflowed


def time_config(name, budget_s=8.0, envs=16):
    """bench.py's NumPy leg: `envs` envs of workload `name` (gym_traffic/workload.py: same prefill, spawn and light
    rules as the GPU run) stepped on ONE core for about budget_s seconds."""
    from gym_traffic import workload as wl
    from gym_traffic.envs.roadgraph import GridRoad
    c = wl.CONFIGS[name]
    envs = max(1, min(envs, c["envs"]))
    g = GridRoad(c["m"], c["n"], c["length"])
    g.generate_entrypoints(0)
    env = NumpyBatchedEnv(c["m"], c["n"], c["length"], c["capacity"], g.dest, g.phases, g.nexts, n_envs=envs)
    env.reset(np.zeros(env.I, np.int32))
    x, v, leading, lastcar = wl.prefill_one_env(c["m"], c["n"], c["length"], c["capacity"], c["prefill"], c["gap"])
    env.x[:], env.v[:], env.leading[:], env.lastcar[:] = x[None], v[None], leading[None], lastcar[None]
    ids = np.arange(envs)
    entry = np.asarray(g.entrypoints)

    def inputs(t):
        cnt = (t % wl.SPAWN_PERIOD == entry % wl.SPAWN_PERIOD).astype(np.int32)
        return wl.cycle_actions(ids, env.I, t), np.tile(cnt[None, :], (envs, 1))
    env.step(*inputs(0), entrypoints=entry)
    base = env.vehicle_updates
    t0 = time.perf_counter()
    ticks = 0
    while time.perf_counter() - t0 < budget_s:
        a, s = inputs(1 + ticks)
        env.step(a, s, entrypoints=entry)
        ticks += 1
    dt = time.perf_counter() - t0
    return {"value": (env.vehicle_updates - base) / dt, "unit": "vehicle-updates/s", "cores": 1, "kind": "port",
            "sample": "%d envs x %d ticks of %s in %.1f s, oracle/numpy_env.py (whole-array NumPy over envs, roads and "
                      "cars; add_car / advance_finished_cars road by road)" % (envs, ticks, name, dt)}
