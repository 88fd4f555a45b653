// tfx_arrivals.cpp - host-side replay of the reference's arrival generators for MANY envs at once.
//
// The reference draws its arrivals from a legacy numpy.random.RandomState per env (traffic_env.py:
// 160-176, :274-283).  gym_traffic/spawner.py replays that call for call in Python, which is fine for
// one env and hopeless for thousands (a few microseconds per draw).  This file restates the three
// legacy routines involved - MT19937's genrand, `random_sample` (53-bit double from two draws),
// `standard_exponential` (-log(1 - u)) and the masked-rejection `randint` on 32-bit draws - so that
// E streams advance in C with the very same bit streams; tests/test_host_logic.py checks them against
// NumPy itself.  Plain C++, no HIP: it is part of libtfx_hip.so only for packaging.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "tfx.h"

namespace {

inline uint32_t mt_next(tfx_arrival_stream *s) {
  constexpr int N = 624, M = 397;
  if (s->pos >= N) {
    uint32_t *mt = s->mt;
    for (int k = 0; k < N; ++k) {
      const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % N] & 0x7fffffffu);
      uint32_t v = mt[(k + M) % N] ^ (y >> 1);
      if (y & 1u) v ^= 0x9908b0dfu;
      mt[k] = v;
    }
    s->pos = 0;
  }
  uint32_t y = s->mt[s->pos++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

inline double mt_double(tfx_arrival_stream *s) {  // legacy random_sample: rk_double
  const uint32_t a = mt_next(s) >> 5, b = mt_next(s) >> 6;
  return (a * 67108864.0 + b) / 9007199254740992.0;
}

inline int32_t mt_randint(tfx_arrival_stream *s, int32_t n) {  // RandomState.randint(0, n), n >= 1
  const uint32_t rng = (uint32_t)(n - 1);
  if (rng == 0) return 0;  // consumes nothing
  uint32_t mask = rng;
  mask |= mask >> 1;
  mask |= mask >> 2;
  mask |= mask >> 4;
  mask |= mask >> 8;
  mask |= mask >> 16;
  for (;;) {
    const uint32_t v = mt_next(s) & mask;
    if (v <= rng) return (int32_t)v;
  }
}

}  // namespace

extern "C" int tfx_arrivals_replay(tfx_arrival_stream *streams, int32_t n_streams, int32_t n_ticks,
                                   int32_t poisson, double mean_gap, int32_t every, int32_t burst,
                                   int32_t n_choices, const int32_t *column_of_choice, int32_t n_columns,
                                   int32_t *counts, int32_t *made) {
  if (!streams || !counts || !column_of_choice || n_streams < 0 || n_ticks < 0 || n_choices < 1 || n_columns < 1)
    return TFX_EINVAL;
  std::memset(counts, 0, (size_t)n_ticks * n_streams * n_columns * sizeof(int32_t));
  // streams are independent: split them over a few host threads (TFX_HOST_THREADS, default 8)
  int n_thr = 8;
  if (const char *ev = std::getenv("TFX_HOST_THREADS")) n_thr = std::atoi(ev);
  const long units = (long)n_streams * n_ticks;  // a thread is worth starting for ~16k stream-ticks
  if (n_thr > units / 16384) n_thr = (int)(units / 16384);
  if (n_thr > n_streams) n_thr = n_streams;
  if (n_thr < 1) n_thr = 1;
  auto work = [&](int32_t k_lo, int32_t k_hi) {
  for (int32_t k = k_lo; k < k_hi; ++k) {
    tfx_arrival_stream *s = &streams[k];
    for (int32_t t = 0; t < n_ticks; ++t) {
      int32_t *row = counts + ((size_t)t * n_streams + k) * n_columns;
      int32_t n = 0;
      if (poisson) {
        // traffic_env.py:160-164: round(Exp(mean gap)) empty ticks, then one car
        for (;;) {
          if (s->gap < 0) s->gap = (int32_t)std::nearbyint(mean_gap * -std::log(1.0 - mt_double(s)));  // round(): half to even
          if (s->gap > 0) {
            --s->gap;
            break;
          }
          s->gap = -1;
          ++row[column_of_choice[mt_randint(s, n_choices)]];  // rand.choice(entrypoints) (:281)
          ++n;
        }
      } else {
        // traffic_env.py:167-176: ceil(cars per tick) cars every round(1 / cars per tick) ticks
        const bool due = every == 0 || s->tick % every == 0;
        ++s->tick;
        if (due)
          for (int32_t b = 0; b < burst; ++b) {
            ++row[column_of_choice[mt_randint(s, n_choices)]];
            ++n;
          }
      }
      if (made) made[(size_t)t * n_streams + k] = n;
    }
  }
  };
  if (n_thr == 1) {
    work(0, n_streams);
  } else {
    std::vector<std::thread> pool;
    for (int i = 0; i < n_thr; ++i)
      pool.emplace_back(work, (int32_t)((long)n_streams * i / n_thr), (int32_t)((long)n_streams * (i + 1) / n_thr));
    for (auto &th : pool) th.join();
  }
  return TFX_OK;
}
