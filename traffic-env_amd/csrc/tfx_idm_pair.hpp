// tfx_idm_pair.hpp - the IDM step for two cars at once in packed fp32 (the walk of k_res, tfx_resident.hpp).
//
// Rounds 1-2 built a streaming kernel around it (k_move_t2: same traffic and results as k_move_t, a third fewer
// vector instructions) and measured it 2-3 % SLOWER at cfg2 three times over - that launch is not bound by its
// instruction count - so the kernel is gone (git history: tfx_move_t2.hpp); k_res, which runs one or two wavefronts
// per SIMD and IS instruction-bound, keeps the arithmetic.  What it does:
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

}  // namespace tfx
