"""Host-side car arrival schedules.

`SpawnSchedule` replays, call for call on a legacy `numpy.random.RandomState`, the two car
generators of the reference and the entry-road draw of `add_new_cars` (traffic_env.py:160-176 and
:274-283), so that `seed_generator(seed)` yields the same (tick, road) sequence as the reference -
"identical seeds/spawns".  RNG draws per car, in order: `exponential` (gap to the next car, Poisson
only), `randint(n_archetypes)`, `choice(entrypoints)`.  The device sees per-tick counts (and, with several archetypes,
the table row of every car).

Two of those calls are issued in a cheaper, stream-identical form (tests/test_host_logic.py checks
both against the literal calls): `choice(a)` of a 1-D array draws `randint(0, len(a))` and indexes,
and `randint(1)` - the archetype pick with the reference's single archetype - consumes nothing.
"""
import math

import numpy as np


class SpawnSchedule(object):
    def __init__(self, rand, poisson, entrypoints, rate_fn, n_archetypes=1):
        """rate_fn() -> (cars_per_sec, rate) is read lazily at the first tick, like the reference's
        generators read FLAGS when first advanced (reset_entrypoints sets cars_per_sec after
        seed_generator in make_env, traffic_test.py:81-82)."""
        self.rand = rand
        self.poisson = bool(poisson)
        self.entrypoints = entrypoints
        self.rate_fn = rate_fn
        self.n_archetypes = int(n_archetypes)
        self._started = False
        self._gap = None      # whole ticks left before the next Poisson car; None = draw a new gap
        self._i = 0           # tick counter of the regular generator

    def _start(self):
        cps, rate = self.rate_fn()
        per_tick = cps * rate
        if self.poisson:
            self._mean_gap = 1 / per_tick
        else:
            self._every = round(1 / per_tick)
            self._burst = math.ceil(per_tick)
        self._started = True

    def _poisson_tick(self):
        n = 0
        while True:
            if self._gap is None:
                self._gap = round(self.rand.exponential(self._mean_gap))
            if self._gap > 0:
                self._gap -= 1
                return n
            self.rows.append(int(self.rand.randint(self.n_archetypes)) if self.n_archetypes > 1 else 0)   # archetype pick
            self._gap = None
            n += 1
            self._roads.append(int(self.entrypoints[self.rand.randint(0, len(self.entrypoints))]))

    def _regular_tick(self):
        due = self._every == 0 or self._i % self._every == 0
        self._i += 1
        if due:
            for _ in range(self._burst):
                self.rows.append(0)           # (the regular generator yields archetypes[0], traffic_env.py:174)
                self._roads.append(int(self.entrypoints[self.rand.randint(0, len(self.entrypoints))]))

    def next_tick(self):
        """Entry roads of the cars created this tick, in creation order (`rows`: the archetype-table row of each)."""
        if not self._started:
            self._start()
        self._roads = []
        self.rows = []
        if self.poisson:
            self._poisson_tick()
        else:
            self._regular_tick()
        return self._roads


def counts_from_roads(roads, entry_index, n_entry, out=None):
    """int32[n_entry] cars per entry road for one tick (cars are identical: only counts matter)."""
    if out is None:
        out = np.zeros(n_entry, np.int32)
    else:
        out[:] = 0
    for rd in roads:
        out[entry_index[rd]] += 1
    return out


class ArrivalStreams(object):
    """E seeded arrival generators advanced together in C (tfx_arrivals_replay, csrc/tfx_arrivals.cpp):
    env k draws exactly what `SpawnSchedule(np.random.RandomState(seeds[k]), ...)` would - the same
    MT19937 words, the same legacy exponential / randint / choice arithmetic - at tens of nanoseconds
    per draw instead of microseconds, so reference-identical arrivals stay affordable for thousands of
    envs.  `next_ticks(n)` returns (counts int32 [n, E, n_columns], made int32 [n, E])."""

    def __init__(self, seeds, poisson, entrypoints, column_of_road, n_columns, cars_per_tick):
        import ctypes as C
        from gym_traffic import _native as nat
        self._C, self._lib = C, nat.lib()

        class Stream(C.Structure):
            _fields_ = [("mt", C.c_uint32 * 624), ("pos", C.c_int32), ("gap", C.c_int32), ("tick", C.c_int64)]
        self.E = len(seeds)
        self._streams = (Stream * self.E)()
        for k, seed in enumerate(seeds):
            st = (seed if isinstance(seed, np.random.RandomState) else np.random.RandomState(seed)).get_state()
            C.memmove(self._streams[k].mt, np.ascontiguousarray(st[1], np.uint32).ctypes.data, 624 * 4)
            self._streams[k].pos, self._streams[k].gap, self._streams[k].tick = int(st[2]), -1, 0
        self.poisson = bool(poisson)
        self.mean_gap = 1 / cars_per_tick
        self.every, self.burst = round(1 / cars_per_tick), math.ceil(cars_per_tick)
        self.columns = np.ascontiguousarray([column_of_road[int(rd)] for rd in entrypoints], np.int32)
        self.n_columns = int(n_columns)
        self._bufs = {}

    def next_ticks(self, n, counts=None, made=None):
        C = self._C
        if counts is None or made is None:      # buffers are kept per n and overwritten by the next call
            buf = self._bufs.get(n)
            if buf is None:
                buf = self._bufs[n] = (np.empty((n, self.E, self.n_columns), np.int32), np.empty((n, self.E), np.int32))
            counts = buf[0] if counts is None else counts
            made = buf[1] if made is None else made
        rc = self._lib.tfx_arrivals_replay(C.cast(self._streams, C.c_void_p), self.E, int(n), int(self.poisson),
                                           float(self.mean_gap), int(self.every), int(self.burst),
                                           int(self.columns.size), self.columns.ctypes.data_as(C.c_void_p),
                                           self.n_columns, counts.ctypes.data_as(C.c_void_p),
                                           made.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise RuntimeError("tfx_arrivals_replay failed (%d)" % rc)
        return counts, made

    def random_state(self, k):
        """The RandomState env k's stream has reached (a copy; for checks and hand-over)."""
        rs = np.random.RandomState(0)
        mt = np.frombuffer(bytes(self._streams[k].mt), np.uint32).copy()
        rs.set_state(('MT19937', mt, int(self._streams[k].pos), 0, 0.0))
        return rs
