"""Host mirror of the on-device arrival generator (csrc/tfx_misc.hpp k_poisson).

The device draws car arrivals with Philox4x32-10 streams keyed by (seed, global env id) and turns
uniforms into whole-tick gaps through a table of 32-bit thresholds built here from the exact
distribution of `round(Exp(mean))` - the reference's gap rule (traffic_env.py:161-163; Python's
round is half-to-even: gap k >= 1 covers [k - 1/2, k + 1/2], with the ties at even k).  Because both
sides compare the same integers, `PoissonMirror` reproduces the device's (tick, road) sequence bit
for bit; tests feed it to the oracle.
"""
import math

import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
TAG_GAP, TAG_ROAD = 0x47415021, 0x524F4144
MASK = 0xFFFFFFFF


def philox4x32(c0, c1, c2, c3, k0, k1):
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c3 ^ k1) & MASK, p0 & MASK
        k0 = (k0 + W0) & MASK
        k1 = (k1 + W1) & MASK
    return c0, c1, c2, c3


def philox4x32_first(c0, c1, c2, c3, k0, k1):
    """First output word of philox4x32 for an array of counters c0 (uint32 values in uint64 arrays)."""
    import numpy as np
    c0 = np.asarray(c0, np.uint64) & np.uint64(MASK)
    c1 = np.full_like(c0, c1)
    c2 = np.full_like(c0, c2)
    c3 = np.full_like(c0, c3)
    m = np.uint64(MASK)
    s32 = np.uint64(32)
    for _ in range(10):
        p0 = np.uint64(M0) * c0
        p1 = np.uint64(M1) * c2
        c0, c1, c2, c3 = ((p1 >> s32) ^ c1 ^ np.uint64(k0)) & m, p1 & m, ((p0 >> s32) ^ c3 ^ np.uint64(k1)) & m, p0 & m
        k0 = (k0 + W0) & MASK
        k1 = (k1 + W1) & MASK
    return c0


def gap_table(cars_per_tick, tail=1e-12):
    """uint32 thresholds cdf[k] = floor(P(gap <= k) * 2^32), last entry 0xFFFFFFFF."""
    mean = 1.0 / float(cars_per_tick)
    cdf = []
    k = 0
    while True:
        p = 1.0 - math.exp(-(k + 0.5) / mean)            # P(Exp(mean) < k + 1/2)
        cdf.append(min(MASK, int(p * 4294967296.0)))
        if 1.0 - p < tail or k > 60000:
            break
        k += 1
    cdf[-1] = MASK
    return np.asarray(cdf, np.uint32)


class PoissonMirror(object):
    """Draw 0 = the first gap; car c uses draw 1 + 2c for its entry road and draw 2 + 2c for the gap
    that follows it (the device evaluates 64 cars at a time from these fixed indices)."""

    def __init__(self, cars_per_tick, seed, n_entry, env_ids):
        self.cdf = [int(c) for c in gap_table(cars_per_tick)]
        self.cdf_np = np.asarray(self.cdf[:-1], np.uint64)
        self.k0, self.k1 = int(seed) & MASK, (int(seed) >> 32) & MASK
        self.n_entry = int(n_entry)
        self.env_ids = [int(e) for e in env_ids]
        self.gap = {e: -1 for e in self.env_ids}
        self.car = {e: 0 for e in self.env_ids}

    def _gap(self, e, draw):
        u = philox4x32(draw & MASK, e, TAG_GAP, 0, self.k0, self.k1)[0]
        k = 0
        while k < len(self.cdf) - 1 and u >= self.cdf[k]:
            k += 1
        return k

    def next_tick(self, frozen=()):
        """int32 [len(env_ids), n_entry] cars per entry road this tick (entry index order)."""
        out = np.zeros((len(self.env_ids), self.n_entry), np.int32)
        for row, e in enumerate(self.env_ids):
            if e in frozen:
                continue
            if self.gap[e] < 0:
                self.gap[e] = self._gap(e, 0)
            if self.gap[e] > 0:
                self.gap[e] -= 1
                continue
            while True:
                # 64 consecutive cars at a time, like the device's wavefront (bursts of thousands of
                # cars per tick at cfg4's rate): every car up to the first non-zero gap arrives now
                c = self.car[e] + np.arange(64, dtype=np.uint64)
                ug = philox4x32_first(np.uint64(2) + np.uint64(2) * c, e, TAG_GAP, 0, self.k0, self.k1)
                gaps = np.searchsorted(self.cdf_np, ug, side='right')     # k with cdf[k-1] <= u < cdf[k]
                stop = np.nonzero(gaps > 0)[0]
                f = int(stop[0]) if stop.size else 63
                ur = philox4x32_first(np.uint64(1) + np.uint64(2) * c[:f + 1], e, TAG_ROAD, 0, self.k0, self.k1)
                np.add.at(out[row], ((ur * np.uint64(self.n_entry)) >> np.uint64(32)).astype(np.int64), 1)
                self.car[e] += f + 1
                if stop.size:
                    self.gap[e] = int(gaps[f]) - 1
                    break
        return out


class RegularMirror(object):
    """Host mirror of the on-device `regular` generator (tfx_set_regular; the reference's traffic_env.py:167-176):
    `burst` = ceil(cars_per_tick) cars in every tick i of an env's generator with i % every == 0, every =
    round(1 / cars_per_tick); car c of env e (counted over the env's whole stream) enters on entry index
    floor(u * n_entry / 2^32) with u the first word of draw 1 + 2c of the env's Philox stream."""

    def __init__(self, cars_per_tick, seed, n_entry, env_ids):
        self.every, self.burst = round(1 / cars_per_tick), math.ceil(cars_per_tick)
        self.k0, self.k1 = int(seed) & MASK, (int(seed) >> 32) & MASK
        self.n_entry = int(n_entry)
        self.env_ids = [int(e) for e in env_ids]
        self.i = {e: 0 for e in self.env_ids}
        self.car = {e: 0 for e in self.env_ids}

    def next_tick(self, frozen=()):
        """int32 [len(env_ids), n_entry] cars per entry road this tick (entry index order)."""
        out = np.zeros((len(self.env_ids), self.n_entry), np.int32)
        for row, e in enumerate(self.env_ids):
            if e in frozen:
                continue
            due = self.every == 0 or self.i[e] % self.every == 0
            self.i[e] += 1
            if due:
                c = self.car[e] + np.arange(self.burst, dtype=np.uint64)
                ur = philox4x32_first(np.uint64(1) + np.uint64(2) * c, e, TAG_ROAD, 0, self.k0, self.k1)
                np.add.at(out[row], ((ur * np.uint64(self.n_entry)) >> np.uint64(32)).astype(np.int64), 1)
                self.car[e] += self.burst
        return out
