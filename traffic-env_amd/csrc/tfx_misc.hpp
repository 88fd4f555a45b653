// tfx_misc.hpp - the cold kernels: reset, tail-cache refresh, remi reward, cars_on_roads, done.
#pragma once
#include "tfx_common.hpp"
#include "tfx_move_t.hpp"

namespace tfx {

// traffic_env.py:259-272
// mask == nullptr: every env; otherwise only the envs whose mask byte is non-zero (tfx_reset_envs)
__global__ void k_reset(const Dev d, const int *phase_init, const uint8_t *mask) {
  const long total = (long)d.E * d.R;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total;
       id += (long)gridDim.x * blockDim.x) {
    const int env = (int)(id / d.R);
    const int e = (int)(id - (long)env * d.R);
    if (mask && !mask[env]) continue;
    if (d.layout == 0) {
      d.xv[(size_t)id * d.C + 1] = make_float2(INFINITY, 0.0f);
      if (d.w) d.w[(size_t)id * d.C + 1] = 0.0f;
    } else {
      d.leadx[id] = INFINITY;
      d.hb[id] = 0;
    }
    d.leading[id] = 1;
    d.lastcar[id] = 1;
    d.tailx[id] = 0.0f;
    d.rec[id] = make_int4(0, 0, 0, 0);
    int *ob = d.obs + (size_t)env * d.obs_len;
    if (e < d.r) {
      ob[e] = 0;
      d.waiting[(size_t)env * d.r + e] = 0;
    }
    if (e < d.I) {
      ob[2 * d.r + e] = phase_init[(size_t)env * d.I + e];
      ob[2 * d.r + d.I + e] = 0;
      d.passed_dst[(size_t)env * d.I + e] = 0;
    }
    if (e == 0) {
      d.done_tick[env] = 0;
      d.env_flag[env] = 0;
      if (d.n_trips) d.n_trips[env] = 0;
    }
  }
}

__global__ void k_refresh(const Dev d) {
  const long total = (long)d.E * d.R;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total;
       id += (long)gridDim.x * blockDim.x)
  {
    if (d.layout == 0) {
      d.tailx[id] = d.xv[(size_t)id * d.C + d.lastcar[id]].x;
    } else {
      const int n = ring_count(d.leading[id], d.lastcar[id], d.C);
      const int hb = d.hb[id];
      d.tailx[id] = (n > 0) ? d.xv[tpos(d, (int)id, n - 1 + hb)].x : 0.0f;
      if (d.het) d.taila[id] = (n > 0) ? side_arch(d.w[tpos(d, (int)id, n - 1 + hb)]) : 0;
    }
  }
}

// transposed layout <-> the reference's ring layout ([E][R][C] (x, v) by ring slot, the fake
// leader's x in slot `leading`): what tfx_export_ring / tfx_import_ring run
__global__ void k_export_ring(const Dev d, float2 *ring, float *ringw, uint8_t *ringa) {
  const long total = (long)d.E * d.R;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total;
       id += (long)gridDim.x * blockDim.x) {
    const int ld = d.leading[id];
    const int n = ring_count(ld, d.lastcar[id], d.C);
    float2 *row = ring + (size_t)id * d.C;
    const int hb = d.layout == 1 ? d.hb[id] : 0;  // rows a two-tick pass left empty at the top of the column (tfx_move_tt.hpp)
    int slot = ld;
    for (int k = 0; k < n; ++k) {
      slot = wrap1(slot + 1, d.C);
      row[slot] = d.xv[tpos(d, (int)id, k + hb)];
      if (d.w && (ringw || ringa)) {  // (heterogeneous cars: the side word is 8 * spawn tick + table row)
        const float sw = d.w[tpos(d, (int)id, k + hb)];
        if (ringw) ringw[(size_t)id * d.C + slot] = side_tick(d, sw);
        if (ringa) ringa[(size_t)id * d.C + slot] = d.het ? (uint8_t)side_arch(sw) : 0;
      }
    }
    row[ld].x = d.leadx[id];
  }
}

__global__ void k_import_ring(const Dev d, const float2 *ring, const float *ringw, const uint8_t *ringa) {
  const long total = (long)d.E * d.R;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total;
       id += (long)gridDim.x * blockDim.x) {
    const int ld = d.leading[id];
    const int n = ring_count(ld, d.lastcar[id], d.C);
    const float2 *row = ring + (size_t)id * d.C;
    int slot = ld;
    for (int k = 0; k < n; ++k) {
      slot = wrap1(slot + 1, d.C);
      d.xv[tpos(d, (int)id, k)] = row[slot];
      if (d.w && (ringw || (d.het && ringa))) {
        const float tk = ringw ? ringw[(size_t)id * d.C + slot] : 0.0f;
        d.w[tpos(d, (int)id, k)] = d.het ? side_pack((int)tk, ringa ? (ringa[(size_t)id * d.C + slot] & (TFX_MAX_ARCH - 1)) : 0) : tk;
      }
    }
    d.leadx[id] = row[ld].x;
    d.hb[id] = 0;  // the column starts at row 0 again
  }
}

// traffic_env.py:64-78
__global__ void k_remi(const Dev d) {
  const long total = (long)d.E * d.I;
  for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
       gid += (long)gridDim.x * blockDim.x) {
    const int env = (int)(gid / d.I);
    const int i = (int)(gid - (long)env * d.I);
    const int cur = d.obs[(size_t)env * d.obs_len + 2 * d.r + i];
    const bool pd = d.passed_dst[gid] != 0;
    float rw = 0.0f;
    for (int dir = 0; dir < 4; ++dir) {
      const int e = dir * d.I + i;
      const int phase_e = (dir < 2) ? 1 : 0;
      const bool green = phase_e != cur;
      int *wp = d.waiting + (size_t)env * d.r + e;
      const bool waiting = *wp > 0;
      if (waiting && !green && !pd) rw -= 0.5f;
      else if (pd && green && !waiting) rw += 0.5f;
      *wp = 0;
    }
    d.rewards[gid] = rw;
    d.passed_dst[gid] = 0;
  }
}

__global__ void k_cars_on_roads(const Dev d, int *out) {
  const long total = (long)d.E * d.R;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total;
       id += (long)gridDim.x * blockDim.x)
    out[id] = ring_count(d.leading[id], d.lastcar[id], d.C);
}

__global__ void k_done(const Dev d, uint8_t *out, int since_tick) {
  for (int env = blockIdx.x * blockDim.x + threadIdx.x; env < d.E; env += gridDim.x * blockDim.x)
    out[env] = d.done_tick[env] > since_tick ? 1 : 0;
}

// ---- on-device arrivals and controller ---------------------------------------
// Philox4x32-10 (Salmon et al., SC'11): counter-based, so env k's stream depends only on
// (seed, global env id, draw index) - identical however the envs are sharded over GPUs.
__device__ __forceinline__ void philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                           unsigned k1, unsigned out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    const unsigned n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    const unsigned n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// The reference's Poisson generator (traffic_env.py:160-164) with a device RNG: one arrival
// process per env; gaps between cars are round(Exp(mean)) whole ticks - drawn here by inverting the
// gap's exact distribution through a host-built table of 32-bit thresholds (cdf[k] = P(gap <= k)
// scaled to 2^32), so host mirror and device agree to the bit without transcendental functions;
// every car picks an entry road uniformly (rand.choice(entrypoints), :280).  Writes this tick's
// car count per entry road.
struct PoissonDev {
  int *counts;             // [E][n_entry]
  int *gap_left;           // [E] whole ticks until the next car; -1 = not drawn yet
  unsigned *draws;         // [E] index of the next car of the env's stream
  const unsigned *cdf;     // [n_cdf] thresholds
  int n_cdf;
  unsigned seed_lo, seed_hi;
  // The reference's `regular` generator instead (traffic_env.py:167-176, tfx_set_regular): `burst` cars in every
  // tick i with i % every == 0 (every tick when every == 0), i = ticks the env's generator has run - kept in
  // gap_left; car c (0-based, counted in draws) takes its entry road from the same draw 1 + 2c as a Poisson car
  int regular, every, burst;
};

// Draw indexing (fixed, so the stream is random-access): draw 0 = the first gap; car c (0-based)
// uses draw 1 + 2c for its entry road and draw 2 + 2c for the gap that follows it.  One WORKGROUP per env
// evaluates blockDim.x consecutive cars at a time: all cars up to and including the first one with a non-zero
// gap arrive in this tick (the reference's generator yields bursts at high rates: thousands of cars per tick
// at cfg4's nominal 15 cars/tick because gaps are rounded to ticks - with one wavefront per env that was 28 us
// per tick, the longest launch of a cfg4 tick; 1024 lanes take the burst in two or three rounds).
// n_ticks > 1: the counts of that many consecutive ticks, rows of E x n_entry each (tfx_step generates a whole
// call's arrivals up front: they depend on nothing but the stream).  Inside an agent step (n_ticks = 1) an env
// that stands still consumes nothing.
__global__ __launch_bounds__(1024) void k_poisson(const Dev d, const PoissonDev ps, const int n_ticks) {
  extern __shared__ int s_hist[];  // [n_entry] + 2 words
  int *s_first = s_hist + d.n_entry;  // lowest car index (relative) whose gap is non-zero | its gap
  const int tid = threadIdx.x, nthr = blockDim.x;
  for (int env = blockIdx.x; env < d.E; env += gridDim.x) {
    const bool frozen = n_ticks == 1 && env_frozen(d, env, *d.tickA);
    int gap = ps.gap_left[env];
    unsigned c0 = ps.draws[env];  // index of the next car
    const unsigned gid = (unsigned)(env + d.env_off);
    unsigned u[4];
    auto gap_of = [&](unsigned draw) {
      philox4x32(draw, gid, 0x47415021u, 0u, ps.seed_lo, ps.seed_hi, u);
      int k = 0;
      while (k < ps.n_cdf - 1 && u[0] >= ps.cdf[k]) ++k;  // cdf[n_cdf-1] catches the tail
      return k;
    };
    __syncthreads();  // (every lane has read the env's state before lane 0 of an earlier iteration's store is overtaken)
    for (int t = 0; t < n_ticks; ++t) {
      for (int j = tid; j < d.n_entry; j += nthr) s_hist[j] = 0;
      __syncthreads();
      if (!frozen && ps.regular) {
        const bool due = ps.every == 0 || gap % ps.every == 0;
        ++gap;
        if (due) {
          for (int j = tid; j < ps.burst; j += nthr) {
            philox4x32(1u + 2u * (c0 + (unsigned)j), gid, 0x524F4144u, 0u, ps.seed_lo, ps.seed_hi, u);
            atomicAdd(&s_hist[(int)(((unsigned long long)u[0] * (unsigned)d.n_entry) >> 32)], 1);
          }
          c0 += (unsigned)ps.burst;
        }
      } else if (!frozen) {
        if (gap < 0) gap = gap_of(0u);
        if (gap > 0) {
          --gap;
        } else {
          for (int guard = 0; guard < 4096; ++guard) {  // (a burst ends with probability 1 - cdf[0] per car)
            if (tid == 0) { s_first[0] = 0x7fffffff; }
            __syncthreads();
            const unsigned c = c0 + (unsigned)tid;
            const int g = gap_of(2u + 2u * c);
            if (g > 0) atomicMin(&s_first[0], tid);
            __syncthreads();
            const int first = s_first[0];
            const int f = first < nthr ? first : nthr - 1;  // last car of this tick within the batch
            if (tid <= f) {
              philox4x32(1u + 2u * c, gid, 0x524F4144u, 0u, ps.seed_lo, ps.seed_hi, u);
              atomicAdd(&s_hist[(int)(((unsigned long long)u[0] * (unsigned)d.n_entry) >> 32)], 1);
            }
            if (tid == f) s_first[1] = g;
            c0 += (unsigned)(f + 1);
            __syncthreads();
            if (first < nthr) {
              gap = s_first[1] - 1;
              break;
            }
          }
        }
      }
      __syncthreads();
      int *row = ps.counts + ((size_t)t * d.E + env) * d.n_entry;
      for (int j = tid; j < d.n_entry; j += nthr) row[j] = s_hist[j];
      __syncthreads();
    }
    if (tid == 0 && !frozen) {
      ps.gap_left[env] = gap;
      ps.draws[env] = c0;
    }
  }
}

// The greedy controller (algorithms/greedy.py:14-16): every `spacing` ticks, phase 1 iff the two
// N-S approaches hold more cars than the two E-W ones (cars_on_roads().dot([1,1,-1,-1]) < 0);
// between decisions the action is held.  One lane per intersection; runs before k_move so the
// counts are the ones an agent would observe before stepping.
// Launched once at the start of a call (the cars may have been edited from outside since the last tick); inside a
// call the advance of tick t leaves the decision for tick t + 1 behind (greedy_decide in advance_item).
__global__ void k_greedy(const Dev d, int *action, int spacing) {
  const int tick = *d.tickA;
  if (tick % spacing != 0) return;
  const long total = (long)d.E * d.I;
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
    const int env = (int)(g / d.I);
    action[g] = greedy_decide(d, env, (int)(g - (long)env * d.I));
  }
}

// start of a fused agent step: remember the tick it starts at (env_frozen compares against it)
__global__ void k_agent_begin(const Dev d, int *first) { *first = *d.tickA; }

// the second half of a split call (tfx_step) keeps clock words of its own: brought up to date at the fork
__global__ void k_clock_copy(const int *tickA, const int *tickB, int *clock2) {
  clock2[0] = *tickA;
  clock2[1] = *tickB;
}

// The end of a fused decision in ONE launch (round 3: four): remi's reward per intersection (traffic_env.py:64-78, as
// k_remi) or the summed rewards as they stand, copied to areward; Repeater's observation (traffic_test.py:48-53): [sum
// of passed | last detected | elapsed/100 * (2*phase - 1)] as float32, from the int obs the ticks left behind (passed
// accumulated in place); the done flags: overflow in any tick since the decision began.  No element depends on another's.
__global__ void k_agent_tail(const Dev d, const int remi, float *aobs, float *areward, uint8_t *adone, const int *first) {
  const int alen = 2 * d.r + d.I;
  const long n_int = (long)d.E * d.I, total = (long)d.E * alen;
  const int f = *first;
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
    if (g < n_int) {
      float rw;
      if (remi) {
        const int env = (int)(g / d.I);
        const int i = (int)(g - (long)env * d.I);
        const int cur = d.obs[(size_t)env * d.obs_len + 2 * d.r + i];
        const bool pd = d.passed_dst[g] != 0;
        rw = 0.0f;
        for (int dir = 0; dir < 4; ++dir) {
          const bool green = ((dir < 2) ? 1 : 0) != cur;
          int *wp = d.waiting + (size_t)env * d.r + dir * d.I + i;
          const bool waiting = *wp > 0;
          if (waiting && !green && !pd) rw -= 0.5f;
          else if (pd && green && !waiting) rw += 0.5f;
          *wp = 0;
        }
        d.rewards[g] = rw;
        d.passed_dst[g] = 0;
      } else {
        rw = d.rewards[g];
      }
      if (areward) areward[g] = rw;
    }
    if (adone && g < d.E) adone[g] = d.done_tick[g] > f ? 1 : 0;
    if (aobs) {
      const int env = (int)(g / alen);
      const int k = (int)(g - (long)env * alen);
      const int *ob = d.obs + (size_t)env * d.obs_len;
      float v;
      if (k < 2 * d.r) {
        v = (float)ob[k];
      } else {
        const int i = k - 2 * d.r;
        v = (float)((double)ob[2 * d.r + d.I + i] / 100.0 * (double)(2 * ob[2 * d.r + i] - 1));
      }
      aobs[g] = v;
    }
  }
}

// Exhaustive check of div_const(a, c, rc) == a / c over every float a with lo <= |a| <= hi (both
// signs) and a == +-0.  Signed zeros compare equal (the consumers add the quotient to another
// value or raise it to the 4th power, so the sign of a zero quotient never survives).
__global__ void k_div_selftest(float c, float rc, float lo, float hi, unsigned long long *mismatches) {
  const unsigned lo_b = __float_as_uint(lo), hi_b = __float_as_uint(hi);
  const unsigned long long n = (unsigned long long)(hi_b - lo_b) + 1ull;
  unsigned long long bad = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n + 2;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    unsigned bits;
    if (i < n) bits = lo_b + (unsigned)i;
    else if (i < 2 * n) bits = (lo_b + (unsigned)(i - n)) | 0x80000000u;
    else bits = (i == 2 * n) ? 0u : 0x80000000u;
    const float a = __uint_as_float(bits);
    const float want = a / c;
    const float got = div_const(a, c, rc);
    const bool same = (__float_as_uint(want) == __float_as_uint(got)) || (want == 0.0f && got == 0.0f);
    bad += same ? 0 : 1;
  }
  if (bad) atomicAdd(mismatches, bad);
}

}  // namespace tfx
