// tfx_misc.hpp - the cold kernels: reset, tail-cache refresh, remi reward, cars_on_roads, done.
#pragma once
#include "tfx_common.hpp"

namespace tfx {

// traffic_env.py:259-272
__global__ void k_reset(const Dev d, const int *phase_init) {
  const long total = (long)d.E * d.R;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total;
       id += (long)gridDim.x * blockDim.x) {
    const int env = (int)(id / d.R);
    const int e = (int)(id - (long)env * d.R);
    d.xv[(size_t)id * d.C + 1] = make_float2(INFINITY, 0.0f);
    if (d.w) d.w[(size_t)id * d.C + 1] = 0.0f;
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
    d.tailx[id] = d.xv[(size_t)id * d.C + d.lastcar[id]].x;
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

// done flags of an agent step: overflow in any tick since the step began
__global__ void k_done_since(const Dev d, uint8_t *out, const int *first) {
  const int f = *first;
  for (int env = blockIdx.x * blockDim.x + threadIdx.x; env < d.E; env += gridDim.x * blockDim.x)
    out[env] = d.done_tick[env] > f ? 1 : 0;
}

// start of a fused agent step: remember the tick it starts at (env_frozen compares against it)
__global__ void k_agent_begin(const Dev d, int *first) { *first = *d.tickA; }

// Repeater's observation (traffic_test.py:48-53): [sum of passed | last detected | elapsed/100 *
// (2*phase - 1)] as float32, from the int obs the ticks left behind (passed accumulated in place).
__global__ void k_agent_obs(const Dev d, float *aobs) {
  const int alen = 2 * d.r + d.I;
  const long total = (long)d.E * alen;
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
    const int env = (int)(g / alen);
    const int k = (int)(g - (long)env * alen);
    const int *ob = d.obs + (size_t)env * d.obs_len;
    float v;
    if (k < 2 * d.r) {
      v = (float)ob[k];
    } else {
      const int i = k - 2 * d.r;
      const int mult = 2 * ob[2 * d.r + i] - 1;
      v = (float)((double)ob[2 * d.r + d.I + i] / 100.0 * (double)mult);
    }
    aobs[g] = v;
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
