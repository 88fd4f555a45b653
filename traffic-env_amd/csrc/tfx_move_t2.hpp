// tfx_move_t2.hpp - k_move_t2: the transposed-layout move kernel with the vector-ALU work cut down.
//
// Same tile ownership, same memory traffic, same stores and bit-identical results as k_move_t
// (tfx_move_t.hpp); what changes is how the arithmetic is issued.  Round-1 counters put k_move_t at
// ~80 vector instructions per 64-car row - 0.40 ms of VALU time next to 0.60 ms of memory time per
// cfg2 launch, close enough that the two did not overlap cleanly.  Here
//   * the walk advances in groups of G rows that are computed TOGETHER: cars k and k+1 of a road are
//     independent under the Jacobi update (both read OLD values), so the ~24 add / mul / fma of two
//     IDM steps issue as v_pk_*_f32 on (car k, car k+1) register pairs;
//   * np.maximum(0, t) and (dx > 0) * dx are single v_max_f32 (exact for every non-NaN operand, see
//     idm_pair), the two constant-divisor divisions stay in reciprocal form;
//   * what makes those forms exact - every v of the group in the self-tested division domain, every
//     gap denominator away from 0 / NaN - is tested ONCE per group with one ballot; a group that
//     fails takes idm_step, the literal form, car by car (never seen in ordinary traffic);
//   * lanes whose road has ended keep computing on zeros instead of being masked off: no EXEC
//     juggling around the arithmetic, only the stores and the counters are predicated;
//   * rows are addressed as a wave-uniform base plus a 32-bit lane offset (no 64-bit multiply-adds).
// Cars spawned this tick continue the chain behind the tail as in k_move_t; their leader's OLD state
// is re-read from the tail row before the walk overwrites it.
#pragma once
#include "tfx_common.hpp"
#include "tfx_move_t.hpp"

namespace tfx {

typedef float v2f __attribute__((ext_vector_type(2)));

// Domain on which idm_pair equals idm_step bit for bit (checked per group by the caller):
//   v, vl  in {+0} U [TFX_FASTDIV_V_LO, TFX_T2_V_HI]   (reciprocal division self-tested there; q^4 and
//                                                        every product below stay finite)
//   |b| >= TFX_T2_B_LO or b = +-inf, b = (xl - x - ll) + eps not NaN   (u = s*/b and u*u finite)
// Then t = v*T + v*(v - vl)/(2 sqrt(ab)), v + dvr and dx are finite, and for finite operands
//   np_max0(t) == v_max_f32(+0, t)   and   x + (dx > 0 ? dx : 0*dx) == x + v_max_f32(dx, -0)
// given the hardware's signed-zero rule max(+0, -0) = +0 in either operand order (checked on the
// device at tfx_create: k_max_selftest; Dev.fastmax).
#define TFX_T2_V_HI 1e4f
#define TFX_T2_B_LO 1e-6f

__device__ __forceinline__ bool t2_v_ok(float v) {
  const unsigned b = __float_as_uint(v);
  const unsigned lo = __float_as_uint(TFX_FASTDIV_V_LO), hi = __float_as_uint(TFX_T2_V_HI);
  return (b == 0u) || ((b - lo) <= (hi - lo));
}

__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f pk_div_const(v2f a, float c, float rc) {
  const v2f q0 = a * rc;
  const v2f r = pk_fma(-q0, (v2f)(c), a);
  return pk_fma(r, (v2f)(rc), q0);
}

// sim (traffic_env.py:50-62) for two cars at once; b = (xl - x - ll) + eps is passed in (the caller
// computed it for the domain test).
__device__ __forceinline__ void idm_pair(const Dev &d, v2f x, v2f v, v2f vl, v2f b, v2f &xn, v2f &vn) {
  const v2f t_gap = v * d.car_T;
  const v2f appr = v * (v - vl);
  const v2f t = t_gap + pk_div_const(appr, d.two_sab, d.r_two_sab);
  v2f m;
  m.x = __builtin_fmaxf(0.0f, t.x);
  m.y = __builtin_fmaxf(0.0f, t.y);
  const v2f s_star = d.car_s0 + m;
  const v2f q = pk_div_const(v, d.car_v0, d.r_v0);
  v2f qd;
  qd.x = pow4_cr(q.x);
  qd.y = pow4_cr(q.y);
  v2f u;
  u.x = s_star.x / b.x;
  u.y = s_star.y / b.y;
  const v2f dv = d.car_a * ((1.0f - qd) - u * u);
  const v2f dvr = dv * d.rate;
  const v2f dx = d.rate * v + (0.5f * dvr) * d.rate;
  v2f adv, vs = v + dvr;
  adv.x = __builtin_fmaxf(dx.x, -0.0f);
  adv.y = __builtin_fmaxf(dx.y, -0.0f);
  xn = x + adv;
  vn.x = __builtin_fmaxf(0.0f, vs.x);
  vn.y = __builtin_fmaxf(0.0f, vs.y);
}

// max(+0, -0) and max(-0, +0) must both be +0, max(x, -0) = x for x > 0 and -0 for x < 0
__global__ void k_max_selftest(unsigned *bad, float pz, float nz) {
  unsigned n = 0;
  n += __float_as_uint(__builtin_fmaxf(pz, nz)) != 0u;
  n += __float_as_uint(__builtin_fmaxf(nz, pz)) != 0u;
  n += __float_as_uint(__builtin_fmaxf(nz, nz)) != 0x80000000u;
  n += __float_as_uint(__builtin_fmaxf(-3.0f * (pz + 1.0f), nz)) != 0x80000000u;
  n += __builtin_fmaxf(2.0f + pz, nz) != 2.0f;
  if (n) atomicAdd(bad, n);
}

template <int G, int NT = 3>
__global__ __launch_bounds__(256) void k_move_t2(const Dev d, const int tidx) {
  static_assert(G >= 2 && (G & 1) == 0, "rows are computed in pairs");
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tick = *d.tickA;
  const int C = d.C;
  const long tiles = (long)d.E * d.G;
  const long nw = (long)gridDim.x * 4;
  const int tick_sp = (d.spawn_mode == TFX_SPAWN_PERIODIC) ? tick % d.spawn_period : 0;
  const unsigned lane8 = (unsigned)lane * 8u;
  const bool fast_ok = d.fastdiv && d.fastmax && !(d.dbg & 64);

  unsigned long long my_updates = 0;

  for (long tile = (long)blockIdx.x * 4 + wv; tile < tiles; tile += nw) {
    const int env = (int)(tile / d.G);  // a tile never straddles envs
    const int e_slot = d.slot_road[(int)(tile - (long)env * d.G) * 64 + lane];
    const bool valid = e_slot >= 0;
    const int e = valid ? e_slot : 0;
    const int id = env * d.R + e;
    const bool run = valid && !env_frozen(d, env, tick);
    const RoadPrep p = prep_road(d, id, env, e, tick, tick_sp, tidx, run, run);
    const int n_old = run ? p.n_old : 0;
    const int n_sp = run ? p.n_tot - p.n_old : 0;

    // wave-uniform tile bases; a row is base + 512 * row + 8 * lane
    char *tb = reinterpret_cast<char *>(d.xv + ((size_t)tile * d.trows) * 64);
    char *ob = reinterpret_cast<char *>(d.outb + ((size_t)tile * KP) * 64);
    auto ldrow = [&](int row) {  // row is wave-uniform
      const f2v t = (NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const f2v *>(tb + (size_t)row * 512 + lane8))
                             : *reinterpret_cast<const f2v *>(tb + (size_t)row * 512 + lane8);
      return t;
    };
    auto strow = [&](unsigned voff, float a, float b) {  // voff = 512 * row + 8 * lane, per lane
      f2v t;
      t.x = a;
      t.y = b;
      if (NT & 2) __builtin_nontemporal_store(t, reinterpret_cast<f2v *>(tb + voff));
      else *reinterpret_cast<f2v *>(tb + voff) = t;
    };

    int kmax = n_old;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int o = __shfl_xor(kmax, off, 64);
      kmax = o > kmax ? o : kmax;
    }
    kmax = __builtin_amdgcn_readfirstlane(kmax);

    // the spawned cars' leader is the tail car's OLD state: read it before the walk overwrites the row
    float sp_lx = p.xL, sp_lv = 0.0f, sp_ll = 0.0f;
    const bool any_spawn = __builtin_amdgcn_ballot_w64(n_sp > 0) != 0ull;
    if (any_spawn && n_sp > 0 && n_old > 0) {
      const f2v t = *reinterpret_cast<const f2v *>(tb + (unsigned)(n_old - 1) * 512u + lane8);
      sp_lx = t.x;
      sp_lv = t.y;
      sp_ll = d.car_l;
    }

    // per-road running state
    float xprev = p.xL, vprev = 0.0f, llv = 0.0f;  // leader of the next car: starts as the fake one
    int kpop = 0, n_wait = 0, n_det = 0;
    bool open = true, far = false;
    float tail_x = 0.0f;
    unsigned wr = lane8;  // where this road's next survivor goes: 512 * (k - kpop) + 8 * lane
    // wrapped ring: x, not v, is tested on slots 1..lastcar (:210) = the cars from index kq on
    const int kq = (p.ld > p.lc) ? C - 1 - p.ld : 0x7fffffff;

    // bookkeeping of one computed car (active lanes only matter; inactive lanes hold zeros)
    auto book = [&](int k, bool active, float xn, float vn) {
      const bool pop = open && active && (xn > d.length) && !(d.dbg & 128);  // the while loop of :123
      open = pop;
      if (pop) {
        if (kpop < KP) {
          *reinterpret_cast<f2v *>(ob + (unsigned)kpop * 512u + lane8) = f2v{xn, vn};
        } else {  // third pop: from here on every car stays in its own row (see k_move_t)
          wr = (unsigned)k * 512u + lane8;
          strow(wr, xn, vn);
          wr += 512u;
        }
        far = far || ((xn - d.length) > d.length);
      } else if (active) {
        strow(wr, xn, vn);  // row k - pops; row k once the road is uncompacted
        wr += 512u;
      }
      kpop += pop ? 1 : 0;
      const float wq = (k >= kq) ? xn : vn;
      n_wait += (active && wq < d.thresh) ? 1 : 0;
      n_det += (active && xn > d.near_end) ? 1 : 0;
      tail_x = active ? xn : tail_x;
    };

    // ---- cars in memory: rows 0 .. kmax-1 in groups of G, the next group in flight -----------------
    f2v pf[G];
#pragma unroll
    for (int u = 0; u < G; ++u) pf[u] = (u < n_old) ? ldrow(u) : f2v{0.0f, 0.0f};
    for (int k0 = 0; k0 < kmax; k0 += G) {
      f2v cur[G];
#pragma unroll
      for (int u = 0; u < G; ++u) cur[u] = pf[u];
      if (k0 + G < kmax) {
#pragma unroll
        for (int u = 0; u < G; ++u) pf[u] = (k0 + G + u < n_old) ? ldrow(k0 + G + u) : f2v{0.0f, 0.0f};
      }
      // leaders: car k-1 of the same road, OLD values (Jacobi); the fake leader / the previous group's
      // last car for the first row
      float xl[G], vl[G], ll[G];
      xl[0] = xprev;
      vl[0] = vprev;
      ll[0] = llv;
#pragma unroll
      for (int u = 1; u < G; ++u) {
        xl[u] = cur[u - 1].x;
        vl[u] = cur[u - 1].y;
        ll[u] = d.car_l;
      }
      // gap denominators b = (x_leader - x - l_leader) + eps, and the domain test of the group
      float bden[G];
      bool ok = t2_v_ok(vprev);
#pragma unroll
      for (int u = 0; u < G; ++u) {
        bden[u] = ((xl[u] - cur[u].x) - ll[u]) + d.eps;
        ok = ok && t2_v_ok(cur[u].y) && (__builtin_fabsf(bden[u]) >= TFX_T2_B_LO);
      }
      float xn[G], vn[G];
      if (fast_ok && __builtin_amdgcn_ballot_w64(!ok) == 0ull) {
#pragma unroll
        for (int u = 0; u < G; u += 2) {
          v2f x2, v2, vl2, b2, xo, vo;
          x2.x = cur[u].x; x2.y = cur[u + 1].x;
          v2.x = cur[u].y; v2.y = cur[u + 1].y;
          vl2.x = vl[u]; vl2.y = vl[u + 1];
          b2.x = bden[u]; b2.y = bden[u + 1];
          idm_pair(d, x2, v2, vl2, b2, xo, vo);
          xn[u] = xo.x; xn[u + 1] = xo.y;
          vn[u] = vo.x; vn[u + 1] = vo.y;
        }
      } else {
#pragma unroll
        for (int u = 0; u < G; ++u) {
          if (d.dbg & 64) { xn[u] = cur[u].x; vn[u] = cur[u].y; }  // timing ablation: no arithmetic
          else idm_step(d, cur[u].x, cur[u].y, xl[u], vl[u], ll[u], xn[u], vn[u]);
        }
      }
      xprev = cur[G - 1].x;  // OLD state leads the next car (Jacobi)
      vprev = cur[G - 1].y;
      llv = d.car_l;
#pragma unroll
      for (int u = 0; u < G; ++u) book(k0 + u, k0 + u < n_old, xn[u], vn[u]);
    }

    // ---- cars spawned this tick (add_car :97-114): they queue behind the tail ------------------
    if (any_spawn) {
      int smax = n_sp;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(smax, off, 64);
        smax = o > smax ? o : smax;
      }
      smax = __builtin_amdgcn_readfirstlane(smax);
      for (int s = 0; s < smax; ++s) {
        const bool act = s < n_sp;
        const float x = act ? spawned_x(d, p.xs0, s) : 0.0f, v = act ? d.car_v : 0.0f;
        float xn1, vn1;
        idm_step(d, x, v, sp_lx, sp_lv, sp_ll, xn1, vn1);
        book(n_old + s, act, xn1, vn1);
        sp_lx = x;
        sp_lv = v;
        sp_ll = d.car_l;
      }
    }

    // ---- phase W -------------------------------------------------------------------------------
    if (run) {
      const int n_tot = p.n_tot;
      if (e < d.r) {
        int *obs = d.obs + (size_t)env * d.obs_len;
        if (n_tot > 0) {
          d.waiting[(size_t)env * d.r + e] += n_wait;
          obs[d.r + e] = n_det;
        }
        obs[e] = (d.agent_mode && tidx > 0) ? obs[e] + kpop : kpop;
        if (kpop > 0) d.passed_dst[(size_t)env * d.I + e % d.I] = 1;
      }
      d.rec[id] = make_int4(rec_pack(kpop, p.ld, C), rec_y(p.ovf_sp, kpop > KP), __float_as_int(tail_x), n_tot);
      if (far || kpop > KP) d.env_flag[env] = tick + 1;
      d.leadx[id] = p.xL;
      my_updates += (unsigned long long)n_tot;
    }
  }

  for (int off = 32; off > 0; off >>= 1) my_updates += __shfl_down(my_updates, off);
  if (lane == 0 && my_updates) veh_add(d.veh, my_updates);
  if (blockIdx.x == 0 && threadIdx.x == 0) *d.tickB = tick;
}

}  // namespace tfx
