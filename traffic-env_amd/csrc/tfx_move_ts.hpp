// tfx_move_ts.hpp - k_move_ts: the transposed-layout move kernel for launches that cannot fill the
// chip with one wavefront per tile (cfg0, cfg1, small RL batches): FOUR wavefronts share a tile.
//
// k_move_t walks a road's cars one after the other, so a launch with fewer tiles than wave slots
// lasts as long as one walk (cfg1 x 1024 envs: 1280 tiles, 36 us for 24 rows).  The Jacobi update
// only needs OLD neighbours, so the walk splits: segment s of a workgroup takes cars
// [s * kseg, (s + 1) * kseg) of every road of the tile, kseg = ceil(longest road / 4).  Each segment
//   1. learns how many cars popped before its range by re-evaluating the head cars until the pop
//      prefix closes (usually one or two rows; any number is handled),
//   2. loads its cars (and the one in front of its first) into registers and computes them,
//   3. workgroup barrier - every read of old rows has happened - then writes the survivors compacted
//      (row k - pops) and the first TFX_KP popped cars to the outbox, exactly as k_move_t does (a
//      road with more than TFX_KP pops is written back uncompacted, every car in its own row),
//   4. the per-road counts meet in LDS and segment 0 writes the road's outputs.
// Same arithmetic, same stores: results are bit-identical to k_move_t (every transposed-layout test
// runs through this kernel whenever the launch is small).
#pragma once
#include "tfx_common.hpp"
#include "tfx_move_t.hpp"

namespace tfx {

// KS cars per segment held in registers: C - 2 <= S * KS; W: spawn-tick plane; S segments (wavefronts) per tile
template <int KS, bool W = false, int S = 4>
__global__ __launch_bounds__(64 * S) void k_move_ts(const Dev d, const int tidx) {
  __shared__ int s_wait[S][64], s_det[S][64], s_kpop[64];
  __shared__ float s_tail[64];
  const int lane = threadIdx.x & 63;
  const int seg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tick = *d.tickA;
  const int C = d.C;
  const long tiles = (long)d.E * d.G;
  const int tick_sp = (d.spawn_mode == TFX_SPAWN_PERIODIC) ? tick % d.spawn_period : 0;
  unsigned long long my_updates = 0;

  for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int env = (int)(tile / d.G);
    const int e_slot = d.slot_road[(int)(tile - (long)env * d.G) * 64 + lane];
    const bool valid = e_slot >= 0;
    const int e = valid ? e_slot : 0;
    const int id = env * d.R + e;
    const bool run = valid && !env_frozen(d, env, tick);
    // every segment evaluates the road's prologue; nobody stores before the barrier (another
    // segment may still have to read lastcar)
    const RoadPrep p = prep_road(d, id, env, e, tick, tick_sp, tidx, run, false);
    const int n_old = run ? p.n_old : 0;
    const int n_tot = run ? p.n_tot : 0;
    float2 *col = d.xv + ((size_t)tile * d.trows) * 64 + lane;
    float2 *ocol = d.outb + ((size_t)tile * KP) * 64 + lane;
    float *wcol = W ? d.w + ((size_t)tile * d.trows) * 64 + lane : nullptr;
    float *owcol = W ? d.outw + ((size_t)tile * KP) * 64 + lane : nullptr;
    // rows the second tick of a pair left empty at the top of the column (tfx_move_tt.hpp: a handle that runs its calls
    // as pairs takes its single ticks here when the launch is small): read past them, write the column compacted
    const int hb = (run && d.hb) ? d.hb[id] : 0;
    const float2 *colr = col + (size_t)hb * 64;
    const float *wcolr = W ? wcol + (size_t)hb * 64 : nullptr;

    int kmax = n_tot;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int o = __shfl_xor(kmax, off, 64);
      kmax = o > kmax ? o : kmax;
    }
    kmax = __builtin_amdgcn_readfirstlane(kmax);
    const int kseg = (kmax + S - 1) / S;
    const int k0 = seg * kseg;
    const int k1 = (k0 + kseg < kmax) ? k0 + kseg : kmax;

    // old state of car k: a row of T, or a car spawned this tick queueing behind the tail (:97-114)
    auto old_car = [&](int k) {
      if (k < n_old) return colr[(size_t)k * 64];
      return make_float2(spawned_x(d, p.xs0, k - n_old), d.car_v);
    };
    auto idm = [&](float x, float v, float xl, float vl, float ll, float &xn, float &vn) {
      const bool off_domain = __builtin_amdgcn_ballot_w64(!idm_fast_domain(v)) != 0ull;
      if (d.fastdiv && !off_domain) idm_step_fast(d, x, v, xl, vl, ll, xn, vn);
      else idm_step(d, x, v, xl, vl, ll, xn, vn);
    };

    bool open = true, far = false;
    int kpop = 0;
    float lx = p.xL, lv = 0.0f, ll = 0.0f;  // leader of car 0: the fake one
    if (seg > 0) {
      // pops among the cars in front of this segment: the while loop of :123 up to car k0
      for (int k = 0;; ++k) {
        const bool act = open && k < k0 && k < n_tot;
        if (__builtin_amdgcn_ballot_w64(act) == 0ull) break;
        if (act) {
          const float2 c = old_car(k);
          float xn, vn;
          idm(c.x, c.y, lx, lv, ll, xn, vn);
          open = xn > d.length;
          kpop += open ? 1 : 0;
          lx = c.x;
          lv = c.y;
          ll = d.car_l;
        }
      }
      if (k0 > 0 && k0 - 1 < n_tot) {  // the car in front of this segment's first one (old state)
        const float2 c = old_car(k0 - 1);
        lx = c.x;
        lv = c.y;
        ll = d.car_l;
      }
    }
    const int kpop0 = kpop;

    float cx[KS], cv[KS], cw[W ? KS : 1];
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      const int k = k0 + u;
      cx[u] = cv[u] = 0.0f;
      if (W) cw[u] = 0.0f;
      if (k < k1 && k < n_tot) {
        const float2 c = old_car(k);
        cx[u] = c.x;
        cv[u] = c.y;
        if (W) cw[u] = (k < n_old) ? wcolr[(size_t)k * 64] : (float)tick;  // spawned this tick: w = tick
      }
    }
    int n_wait = 0, n_det = 0;
    unsigned long long popmask = 0ull;
    float tail_x = 0.0f;
    bool has_tail = false;
    // wrapped ring: x, not v, is tested on slots 1..lastcar (:210) = the cars from index kq on
    const int kq = (p.ld > p.lc) ? C - 1 - p.ld : 0x7fffffff;
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      const int k = k0 + u;
      if (k < k1 && k < n_tot) {
        const float x = cx[u], v = cv[u];
        float xn, vn;
        idm(x, v, lx, lv, ll, xn, vn);
        cx[u] = xn;
        cv[u] = vn;
        lx = x;  // OLD state leads the next car (Jacobi)
        lv = v;
        ll = d.car_l;
        const bool pop = open && (xn > d.length);
        open = pop;
        if (pop) popmask |= 1ull << u;
        kpop += pop ? 1 : 0;
        far = far || (pop && ((xn - d.length) > d.length));
        const float wq = (k >= kq) ? xn : vn;
        n_wait += (wq < d.thresh) ? 1 : 0;
        n_det += (xn > d.near_end) ? 1 : 0;
        if (k == n_tot - 1) {
          tail_x = xn;
          has_tail = true;
        }
      }
    }
    if (seg == S - 1) s_kpop[lane] = kpop;  // the last segment has seen every car in front of it
    __syncthreads();  // every segment has read the old rows it needs; the road's pop count is known
    if (seg == 0 && hb) d.hb[id] = 0;

    {
      const bool unc = s_kpop[lane] > KP;  // > TFX_KP pops: every car back to its own row (see k_move_t)
      int kp = kpop0;
#pragma unroll
      for (int u = 0; u < KS; ++u) {
        const int k = k0 + u;
        if (k < k1 && k < n_tot) {
          if ((popmask >> u) & 1ull) {
            if (kp < KP) {     // the road's outbox column
              ocol[(size_t)kp * 64] = make_float2(cx[u], cv[u]);
              if (W) owcol[(size_t)kp * 64] = cw[u];
            } else {           // (uncompacted road) later pops stay in their own rows
              col[(size_t)k * 64] = make_float2(cx[u], cv[u]);
              if (W) wcol[(size_t)k * 64] = cw[u];
            }
            ++kp;
          } else {
            const int row = unc ? k : k - kp;
            col[(size_t)row * 64] = make_float2(cx[u], cv[u]);
            if (W) wcol[(size_t)row * 64] = cw[u];
          }
        }
      }
    }
    s_wait[seg][lane] = n_wait;
    s_det[seg][lane] = n_det;
    if (has_tail || (seg == 0 && n_tot == 0)) s_tail[lane] = tail_x;
    if (far || kpop > KP) d.env_flag[env] = tick + 1;
    __syncthreads();

    if (seg == 0 && run) {
      int tot_wait = 0, tot_det = 0;
#pragma unroll
      for (int q = 0; q < S; ++q) {
        tot_wait += s_wait[q][lane];
        tot_det += s_det[q][lane];
      }
      const int kpop_all = s_kpop[lane];
      if (p.n_tot != p.n_old) d.lastcar[id] = p.lc;
      if (e < d.r) {
        int *ob = d.obs + (size_t)env * d.obs_len;
        if (n_tot > 0) {
          d.waiting[(size_t)env * d.r + e] += tot_wait;
          ob[d.r + e] = tot_det;
        }
        ob[e] = (d.agent_mode && tidx > 0) ? ob[e] + kpop_all : kpop_all;
        if (kpop_all > 0) d.passed_dst[(size_t)env * d.I + e % d.I] = 1;
      }
      d.rec[id] = make_int4(rec_pack(kpop_all, p.ld, C), rec_y(p.ovf_sp, kpop_all > KP), __float_as_int(s_tail[lane]), n_tot);
      d.leadx[id] = p.xL;
      my_updates += (unsigned long long)n_tot;
    }
    __syncthreads();  // the LDS words are free for the next tile
  }

  if (seg == 0) {
    for (int off = 32; off > 0; off >>= 1) my_updates += __shfl_down(my_updates, off);
    if (lane == 0 && my_updates) veh_add(d.veh, my_updates);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *d.tickB = tick;
}

}  // namespace tfx
