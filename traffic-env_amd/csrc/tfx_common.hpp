// tfx_common.hpp - device parameter block and the arithmetic every kernel shares.
//
// Float contract (bit-for-bit shared with oracle/idm_oracle.c): binary32, the reference's
// expression order (gym_traffic/envs/traffic_env.py:50-62), one rounding per operation
// (-ffp-contract=off), correctly rounded divide, q^4 from two binary64 multiplies,
// np.maximum(0, t) as (0 >= t ? 0 : t), (dx > 0) * dx as (dx > 0 ? dx : 0 * dx).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "tfx.h"
#include "tfx_pow.h"

namespace tfx {

constexpr int KP = TFX_KP;

// Everything a kernel needs, passed by value.
struct Dev {
  int I, r, R, C, E, n_entry, obs_len;
  int yellow, learn_switch, validate, env_off;
  // fused agent step (Repeater + Remi, traffic_test.py:27-64): `passed` (and, without Remi, the
  // rewards) accumulate over the ticks of the step; an env that overflowed in an earlier tick of
  // the step stands still for the rest of it (`if done: break`, traffic_test.py:55)
  int agent_mode, accum_rewards;
  const int *agent_first;  // tick at which the current agent step began
  float length, rate, car_v, car_l, car_a, car_v0, car_b, car_T, car_s0;
  float risk_a;                  // the largest acceleration any car can have (k_risk's movement bound): car_a, or the table's maximum
  float two_sab, eps, thresh, near_end, ovf_pen;
  float r_two_sab, r_v0;  // correctly rounded reciprocals of the two constant divisors
  int fastdiv;            // the reciprocal form of those divisions was verified exact (div_selftest)
  int fastmax;            // v_max_f32 orders +0 above -0 in either operand order (k_max_selftest)
  // caller-owned state
  float2 *xv;  // [E][R][C] (x, v) per ring slot
  float *w;    // [E][R][C] spawn tick per ring slot, or nullptr
  int *leading, *lastcar, *obs;
  // phase | elapsed of env k's intersections: lights + k * lights_stride (= obs + 2r, obs_len: the words inside obs
  // - or a workgroup's LDS copy with stride 0, k_tail)
  int *lights;
  int lights_stride;
  float *rewards;
  int *waiting;
  uint8_t *passed_dst;
  int *done_tick;
  float *trip_times;
  int *n_trips;
  int trip_cap;
  // handle-owned tables + scratch
  const int *nexts, *pred, *entry_idx;
  // transposed layout: road e of an env lives in storage slot road_slot[e] (tile = slot / 64, lane =
  // slot % 64, G tiles per env); slot_road is the inverse (-1 = padding lane)
  const int *road_slot, *slot_road;
  int G;
  int trows;  // rows (of 64 (x, v) pairs) a tile occupies in T: >= C - 2
  int4 *rec;      // per road: {pops k | head slot << 16, spawn overflows | uncompacted << 30, bits of post-move tail x, live cars}
  // transposed layout: rows at the top of a road's column that hold no car (0..TFX_KP; see rec_hb) - a byte per road
  // of its own since round 4: the pass reads one byte instead of a 16-byte record, k_tail writes one back
  uint8_t *hb;
  // k_tail<AGENT> only (LDS): 0 = the road's head cannot reach the road's end in the next tick - what edge_tile saw while
  // it held the head's new state - so that the bound for the next pair (risk_lane) need not load the road's first rows
  uint8_t *riskhint;
  // "road state word": leading | lastcar << 9 | hb << 18 of a road, 4 bytes.  Between the pairs of ONE tfx_step call
  // (plain cars, outside agent steps) k_tail writes this word instead of leading, lastcar and hb (9 bytes) and the next
  // pass reads it instead of them (its RSW form); the call's last k_tail brings the reference's arrays up to date, its
  // first pass reads those - nobody else looks at them in between.
  int *rsw;
  // per road, two-tick pass only (tfx_move_tt.hpp): what the pass hands to the edge work of the second tick, 12 bytes
  // (round 3: one 16-byte record): {v of the tail after the first tick, x of the tail after the second} and the
  // waiting (both ticks so far) | detected (second tick so far) | detected (first tick) counts (rec2c_pack)
  float2 *rec2f;
  int *rec2c;
  // A two-tick pass that k_tail follows (its CREC form) writes its road record in 8 bytes instead of rec's 16 - {packed
  // integers (crec_pack), bits of the tail's x} - and the count of spawn overflows, when there are any, to ovf_cnt; k_tail expands it into its LDS copy
  // of rec, so the phase code reads what it always read
  int2 *crec;
  int *ovf_cnt;
  // transposed layout only (tfx_config.layout = 1): xv is T[tile][k][64]; the first TFX_KP = 2 cars that
  // left a road this tick wait in its outbox column outb[tile][j][64], j < KP - a road that pops more
  // stays uncompacted for the tick and its env takes the serial advance; the fake leader's x has no
  // slot of its own and lives in leadx
  float2 *outb;
  float *outw;  // outbox of the spawn-tick plane (planes = 3)
  float *leadx;
  int layout;
  float *tailx;   // per road: x of the last car after the advance (what update_lights reads)
  int *env_flag;  // == tick+1 when the env must take the serial advance this tick
  // == tick+1 when the env takes the pair of ticks starting at `tick` one tick at a time (k_risk).  Two planes, used in
  // turn by the pairs of a call (risk_word): k_tail stamps the NEXT pair's plane while the launches behind it still read
  // the current pair's
  int *env_risk;
  int risk_stride;  // words from one plane to the other (the whole handle's E, also inside the half of a split call)
  int *risk_any;  // [2], one per plane of env_risk: == tick+1 when any env is marked for the pair starting at `tick`
  unsigned long long *veh;
  unsigned long long *slow_pairs;  // env-pairs of agent steps that k_tail took one tick at a time (tfx_slow_pairs)
  int *tickA, *tickB;
  // per-tick inputs
  const int *action;
  int action_mode, action_period;
  long action_stride;
  const int *spawn;
  int spawn_mode, spawn_period;
  long spawn_stride;
  // greedy controller (algorithms/greedy.py:14-16): > 0 = the advance of tick t writes the action of tick t + 1
  // whenever (t + 1) % greedy_spacing == 0
  int greedy_spacing;
  int *greedy_act;               // [E][I]
  // heterogeneous cars (tfx_config.n_archetypes): every car's side word w holds (spawn tick mod 2^24) << 6 | table row as integer bits
  int het;
  const float *arch_tab;         // [TFX_MAX_ARCH][ARCH_W]: l, a, v0, T, s0, 2 sqrt(a b), delta, spawn speed
  int *taila;                    // per road: table row of its last car (next to tailx)
  const uint8_t *spawn_arch;     // rows of the cars the count buffer adds: [tick][E][n_entry][spawn_arch_S]
  int spawn_arch_S;
  long spawn_arch_stride;
};

// The vehicle-update counter is spread over VEH_SLOTS words, one cache line apart: every wavefront
// adds its share once, and with one word a small launch (one tile per wave, all waves finishing
// together) spent ~25 us serialising 1280 atomics on it (cfg1 x 1024: k_move_t 30 us -> 5 us + walk).
constexpr int VEH_SLOTS = 256, VEH_STRIDE = 8;
__device__ __forceinline__ void veh_add(unsigned long long *veh, unsigned long long v) {
  const unsigned slot = (blockIdx.x * 4u + (threadIdx.x >> 6)) & (unsigned)(VEH_SLOTS - 1);
  atomicAdd(veh + (size_t)slot * VEH_STRIDE, v);
}

__device__ __forceinline__ float np_max0(float t) { return (0.0f >= t) ? 0.0f : t; }
// (Round 3 priced the two binary64 multiplies + two conversions: with (q*q)*(q*q) in binary32 instead - NOT this
// contract's value - a two-tick pass at cfg2 takes 0.736 ms instead of 0.764.  3.5 % is the most any binary32
// reformulation could win, and one that reproduces RN32(RN64(q^4)) for every q needs the error terms of both squarings
// (>= 9 dependent binary32 operations against these 4), so the binary64 form stays.)
__device__ __forceinline__ float pow4_cr(float q) {
  const double q2 = (double)q * (double)q;
  return (float)(q2 * q2);
}
// traffic_env.py:46-47
__device__ __forceinline__ int wrap1(int a, int C) { return a >= C ? 1 : a; }
// slot reached from `slot` (1..C-1) after k (0..C-1) ring steps
__device__ __forceinline__ int ring_adv(int slot, int k, int C) {
  const int s = slot + k;
  return s >= C ? s - (C - 1) : s;
}
// traffic_env.py:214-218
__device__ __forceinline__ int ring_count(int ld, int lc, int C) { return lc - ld + (ld > lc ? C - 1 : 0); }

// sim (traffic_env.py:50-62) for one car: (x, v) against its leader (xl, vl, length ll).
__device__ __forceinline__ void idm_step(const Dev &d, float x, float v, float xl, float vl, float ll,
                                         float &xn, float &vn) {
  const float t_gap = v * d.car_T;
  const float appr = v * (v - vl);
  const float s_star = d.car_s0 + np_max0(t_gap + appr / d.two_sab);
  const float s = (xl - x) - ll;
  const float q = v / d.car_v0;
  const float qd = pow4_cr(q);
  const float u = s_star / (s + d.eps);
  const float dv = d.car_a * ((1.0f - qd) - u * u);
  const float dvr = dv * d.rate;
  const float dx = d.rate * v + (0.5f * dvr) * d.rate;
  xn = x + (dx > 0.0f ? dx : 0.0f * dx);
  vn = np_max0(v + dvr);
}

// ---- heterogeneous cars ---------------------------------------------------------------------------------
constexpr int ARCH_W = 8;
enum { AR_L = 0, AR_A, AR_V0, AR_T, AR_S0, AR_2SAB, AR_DELTA, AR_V };
// The side word of a heterogeneous car is an INTEGER bit pattern kept in the float plane: (spawn tick mod 2^24) << 6 |
// table row.  (Round 3 stored the float value 8 * tick + row: exact only below 2^24, so from tick 2^21 on the row bits
// were rounded away and new cars got wrong parameters.)  The pattern never has an all-ones exponent (bit 30 is clear),
// loads, stores, moves and selects keep denormal patterns as they are, and no arithmetic ever touches the word.  Trip
// times need only tick differences, taken modulo 2^24 (side_age) - they stay right however long the handle runs.
constexpr int ARCH_BITS = 6;
static_assert((1 << ARCH_BITS) == TFX_MAX_ARCH, "the side word holds the table row in its low bits");
constexpr int SIDE_TICK_MASK = 0xffffff;
__device__ __forceinline__ int side_arch(float wa) { return __float_as_int(wa) & (TFX_MAX_ARCH - 1); }
__device__ __forceinline__ float side_tick(const Dev &d, float wa) {
  return d.het ? (float)(__float_as_int(wa) >> ARCH_BITS) : wa;
}
__device__ __forceinline__ float side_pack(int tick, int row) {
  return __int_as_float(((tick & SIDE_TICK_MASK) << ARCH_BITS) | row);
}
// tick - spawn tick of a car leaving the map (advance_hack, traffic_env.py:154)
__device__ __forceinline__ float side_age(const Dev &d, int tick, float wa) {
  return d.het ? (float)((tick - (__float_as_int(wa) >> ARCH_BITS)) & SIDE_TICK_MASK) : (float)tick - wa;
}
// the archetype table (rows past the handle's own are zero) into a workgroup's LDS copy
__device__ __forceinline__ void load_arch(const Dev &d, float *s_arch) {
  for (int i = threadIdx.x; i < TFX_MAX_ARCH * ARCH_W; i += blockDim.x) s_arch[i] = d.arch_tab[i];
  __syncthreads();
}

// (v/v0)**delta for an integer delta in 1..8: oracle/idm_oracle.c powi_cr, multiply for multiply
__device__ __forceinline__ float powi_cr(float q, int n) {
  double result = 1.0, base = (double)q;
  while (n) {
    if (n & 1) result *= base;
    base *= base;
    n >>= 1;
  }
  return (float)result;
}

// sim (traffic_env.py:50-62) for one car with its OWN parameters (row `me` of the table) behind a leader of
// length ll: the literal expression order of idm_step, IEEE divisions throughout
__device__ __forceinline__ void idm_step_het(const Dev &d, const float *me, float x, float v, float xl, float vl, float ll,
                                             float &xn, float &vn) {
  const float t_gap = v * me[AR_T];
  const float appr = v * (v - vl);
  const float s_star = me[AR_S0] + np_max0(t_gap + appr / me[AR_2SAB]);
  const float s = (xl - x) - ll;
  const float q = v / me[AR_V0];
  // (v/v0)**delta: integer exponents 1..8 by their multiply chains (4: the single-archetype path's pow4_cr), anything
  // else by the binary64 sequence the oracle shares (include/tfx_pow.h)
  const float df = me[AR_DELTA];
  const int delta = (int)df;
  const float qd = (df == (float)delta && delta >= 1 && delta <= 8) ? (delta == 4 ? pow4_cr(q) : powi_cr(q, delta))
                                                                    : tfx_pow_det(q, df);
  const float u = s_star / (s + d.eps);
  const float dv = me[AR_A] * ((1.0f - qd) - u * u);
  const float dvr = dv * d.rate;
  const float dx = d.rate * v + (0.5f * dvr) * d.rate;
  xn = x + (dx > 0.0f ? dx : 0.0f * dx);
  vn = np_max0(v + dvr);
}

// a / c for a constant c with rc = RN(1/c): q0 = RN(a*rc), r = a - q0*c (exact in one FMA),
// q = RN(q0 + r*rc).  For the two constants of the IDM (2*sqrt(a*b) and v0) this equals the
// correctly rounded quotient for EVERY numerator in the domain idm_fast_domain() admits - checked
// exhaustively on the device for the handle's constants (k_div_selftest) before `fastdiv` is set.
// Saves two v_rcp_f32 and ~14 VALU operations per car.
__device__ __forceinline__ float div_const(float a, float c, float rc) {
  const float q0 = a * rc;
  const float r = __builtin_fmaf(-q0, c, a);
  return __builtin_fmaf(r, rc, q0);
}

// v for which both constant-divisor quotients of idm_step are covered by the self-test: v == 0, or
// 1e-10 <= v <= 1e18 (then appr = v*(v - vl) is 0 or 6e-28 <= |appr| <= 2e36 for any admitted vl).
#define TFX_FASTDIV_V_LO 1e-10f
#define TFX_FASTDIV_V_HI 1e18f
#define TFX_FASTDIV_A_LO 5e-28f
#define TFX_FASTDIV_A_HI 4e36f
__device__ __forceinline__ bool idm_fast_domain(float v) {
  // +0, or TFX_FASTDIV_V_LO <= v <= TFX_FASTDIV_V_HI, as ONE unsigned range test on the bit pattern
  // (non-negative floats order like their bits; -0, negatives and NaNs fall outside and take the
  // plain-division path, which is exact everywhere)
  const unsigned b = __float_as_uint(v);
  const unsigned lo = __float_as_uint(TFX_FASTDIV_V_LO), hi = __float_as_uint(TFX_FASTDIV_V_HI);
  return (b == 0u) || ((b - lo) <= (hi - lo));
}

// idm_step with the two constant-divisor divisions in reciprocal form (bit-identical on the
// admitted domain; callers check idm_fast_domain for every live lane first).
__device__ __forceinline__ void idm_step_fast(const Dev &d, float x, float v, float xl, float vl, float ll,
                                              float &xn, float &vn) {
  const float t_gap = v * d.car_T;
  const float appr = v * (v - vl);
  const float s_star = d.car_s0 + np_max0(t_gap + div_const(appr, d.two_sab, d.r_two_sab));
  const float s = (xl - x) - ll;
  const float q = div_const(v, d.car_v0, d.r_v0);
  const float qd = pow4_cr(q);
  const float u = s_star / (s + d.eps);
  const float dv = d.car_a * ((1.0f - qd) - u * u);
  const float dvr = dv * d.rate;
  const float dx = d.rate * v + (0.5f * dvr) * d.rate;
  xn = x + (dx > 0.0f ? dx : 0.0f * dx);
  vn = np_max0(v + dvr);
}

// the stamp of env for the pair that tick index `tidx` of the call belongs to (ticks 2p and 2p + 1: plane p & 1)
__device__ __forceinline__ int &risk_word(const Dev &d, int env, int tidx) {
  return d.env_risk[(size_t)((tidx >> 1) & 1) * d.risk_stride + env];
}

__device__ __forceinline__ int &risk_any_word(const Dev &d, int tidx) { return d.risk_any[(tidx >> 1) & 1]; }

// env stopped for the rest of the current agent step: it overflowed in one of the step's earlier
// ticks (done_tick holds overflow tick + 1)
__device__ __forceinline__ bool env_frozen(const Dev &d, int env, int tick) {
  if (!d.agent_mode) return false;
  const int dt = d.done_tick[env];
  return dt > *d.agent_first && dt <= tick;
}

// TrafficEnv._step lines :225-232 for one intersection: new phase and elapsed from the old ones.
__device__ __forceinline__ void light_next(const Dev &d, int env, int i, int tick, int tidx, int ph, int el,
                                           int &ph_new, int &el_new) {
  int a;
  if (d.action_mode == TFX_ACTION_CYCLE)
    a = ((tick + (env + d.env_off) % d.action_period) / d.action_period) & 1;
  else if (d.action_mode == TFX_ACTION_BROADCAST)
    a = d.action[(size_t)tidx * d.action_stride + i];
  else
    a = d.action[(size_t)tidx * d.action_stride + (size_t)env * d.I + i];
  int change;
  if (d.learn_switch) {
    change = a != 0;
    ph_new = ((ph != 0) != (a != 0)) ? 1 : 0;
  } else {
    change = (ph != 0) != (a != 0);
    ph_new = a;
  }
  el_new = change ? 0 : el + 1;
}
__device__ __forceinline__ void light_update(const Dev &d, int env, int i, int tick, int tidx,
                                             int &ph_new, int &el_new) {
  const int *ob = d.lights + (size_t)env * d.lights_stride;
  light_next(d, env, i, tick, tidx, ob[i], ob[d.I + i], ph_new, el_new);
}

// cars add_new_cars (traffic_env.py:274-283) puts on entry road e (entry index ej) this tick
__device__ __forceinline__ int spawn_count(const Dev &d, int env, int e, int ej, int tick_mod_period, int tidx) {
  if (d.spawn_mode == TFX_SPAWN_COUNTS)
    return d.spawn[(size_t)tidx * d.spawn_stride + (size_t)env * d.n_entry + ej];
  if (d.spawn_mode == TFX_SPAWN_PERIODIC) return (tick_mod_period == e % d.spawn_period) ? 1 : 0;
  return 0;
}

// Phase-M work shared by the move kernels: everything about road (env, e) that does not need the
// cars - ring indices, the fake leader's x from the light state (update_lights :81-94) and the
// spawn pushes (add_car :97-114).  `valid` lanes evaluate the spawns; the lane with `store` set
// also writes the new lastcar (several lanes may prepare the same road redundantly).
struct RoadPrep {
  int ld, lc, n_old, n_tot, ovf_sp;
  float xL, xs0;  // fake-leader x; x of the first car spawned this tick
  int hb;         // RSW form: rows that hold no car at the top of the road's column (otherwise the caller reads d.hb)
};

__device__ __forceinline__ int rsw_pack(int ld, int lc, int hb) { return ld | (lc << 9) | (hb << 18); }

// RSW: the ring indices come from the road state words (Dev::rsw) and a spawn's new lastcar is not stored (k_tail, which
// follows, takes it from the pass's record)
template <bool RSW = false>
__device__ __forceinline__ RoadPrep prep_road(const Dev &d, int id, int env, int e, int tick,
                                              int tick_mod_period, int tidx, bool valid, bool store) {
  const int C = d.C;
  RoadPrep p;
  if (RSW) {
    const int w = d.rsw[id];
    p.ld = w & 511;
    p.lc = (w >> 9) & 511;
    p.hb = (w >> 18) & 3;
  } else {
    p.ld = d.leading[id];
    p.lc = d.lastcar[id];
    p.hb = 0;
  }
  p.n_old = ring_count(p.ld, p.lc, C);
  p.n_tot = p.n_old;
  p.ovf_sp = 0;
  p.xL = INFINITY;
  p.xs0 = 0.0f;
  if (e < d.r) {
    const int dir = e / d.I;
    int ph_new, el_new;
    light_update(d, env, e - dir * d.I, tick, tidx, ph_new, el_new);
    const int phase_e = (dir < 2) ? 1 : 0;  // roadgraph.py:36
    if (phase_e == ph_new || el_new < d.yellow) {
      p.xL = d.length;
    } else {
      const int idn = env * d.R + d.nexts[e];
      bool occupied;
      if (RSW) {
        const int wn = d.rsw[idn];
        occupied = ((wn >> 9) & 511) != (wn & 511);
      } else {
        occupied = d.lastcar[idn] != d.leading[idn];
      }
      if (occupied) p.xL = d.tailx[idn] + d.length;
    }
  }
  const int ej = d.entry_idx[e];
  if (ej >= 0 && valid) {
    const int c = spawn_count(d, env, e, ej, tick_mod_period, tidx);
    if (c > 0) {
      float tail_x = d.tailx[id];
      for (int q = 0; q < c; ++q) {
        const int pos = wrap1(p.lc + 1, C);
        const float start = (p.lc != p.ld) ? (tail_x - d.car_l) - d.car_s0 : INFINITY;
        if (pos != p.ld) {
          const float xv = (start < 0.0f) ? start : 0.0f;  // min(car.x = 0, start)
          if (p.n_tot == p.n_old) p.xs0 = xv;
          ++p.n_tot;
          p.lc = pos;
          tail_x = xv;
        } else {
          ++p.ovf_sp;
        }
      }
      if (!RSW && store && p.n_tot != p.n_old) d.lastcar[id] = p.lc;
    }
  }
  return p;
}

// rec.x of a road: number of cars popped this tick (they sit in ring slots head, head+1, ...)
__device__ __forceinline__ int rec_pack(int kpop, int ld, int C) { return kpop | (wrap1(ld + 1, C) << 16); }
__device__ __forceinline__ int rec_kpop(int rx) { return rx & 0xffff; }
// rec.y: spawn overflows of the tick | "the road was left uncompacted" (transposed layout, > TFX_KP pops)
__device__ __forceinline__ int rec_y(int ovf_sp, bool unc) { return ovf_sp | (unc ? (1 << 30) : 0); }
__device__ __forceinline__ int rec_ovf_sp(int ry) { return ry & 0x0fffffff; }
__device__ __forceinline__ bool rec_unc(int ry) { return (ry >> 30) & 1; }
// d.hb[road]: rows at the top of the road's column that hold no car (transposed layout).  Only the second tick of a
// two-tick pass leaves any (edge_tile: it pops without compacting); the next move kernel reads past them and writes the
// column compacted again (tfx_move_tt.hpp).
__device__ __forceinline__ int rec_head(int rx) { return rx >> 16; }
// rec.w: cars on the road during the move (incl. this tick's arrivals) | table row of the last of them << 16
__device__ __forceinline__ int rec_ntot(int rw) { return rw & 0xffff; }
__device__ __forceinline__ int rec_taila(int rw) { return rw >> 16; }

// rec2c: waiting count of both ticks so far (10 bits) | detected of the second tick so far (9) << 10 | detected of the
// first tick (9) << 19 | the road held cars in the first tick << 28
__device__ __forceinline__ int rec2c_pack(int wait, int det1, int det0, bool had_cars) {
  return wait | (det1 << 10) | (det0 << 19) | (had_cars ? 1 << 28 : 0);
}
// Two forms of crec.x.  Heterogeneous cars: pops | cars on the road << 9 | table row of the tail << 18 | uncompacted << 24
// | spawn overflows? << 25.  Otherwise the cars on the road follow from the ring indices, and those ride along instead -
// pops | leading << 9 | lastcar << 18 (ring slots: <= 257) | uncompacted << 27 | spawn overflows? << 28 - so that k_tail
// need not load leading / lastcar (tfx_step calls; inside agent steps an env may have been skipped by the pass, and
// k_tail loads them).
template <bool HET>
__device__ __forceinline__ int crec_pack(int kpop, int n_tot, int ld, int lc, int taila, bool unc, bool ovf) {
  if (HET) return kpop | (n_tot << 9) | (taila << 18) | (unc ? 1 << 24 : 0) | (ovf ? 1 << 25 : 0);
  return kpop | (ld << 9) | (lc << 18) | (unc ? 1 << 27 : 0) | (ovf ? 1 << 28 : 0);
}
template <bool HET>
__device__ __forceinline__ bool crec_ovf(int cx) { return (cx >> (HET ? 25 : 28)) & 1; }
// the 16-byte road record from its 8-byte form (ovf_sp: the spawn overflows, read from ovf_cnt when crec_ovf says so;
// the head slot of rec.x is a ring-layout notion and stays 0)
template <bool HET>
__device__ __forceinline__ int4 crec_expand(int2 c, int ovf_sp, int C) {
  if (HET) return make_int4(c.x & 511, rec_y(ovf_sp, (c.x >> 24) & 1), c.y, ((c.x >> 9) & 511) | (((c.x >> 18) & 63) << 16));
  return make_int4(c.x & 511, rec_y(ovf_sp, (c.x >> 27) & 1), c.y, ring_count((c.x >> 9) & 511, (c.x >> 18) & 511, C));
}

// greedy.py:14-16: phase 1 iff the two N-S approaches hold more cars than the two E-W ones
// (cars_on_roads().dot([1,1,-1,-1]) < 0) at intersection i of env
__device__ __forceinline__ int greedy_decide(const Dev &d, int env, int i) {
  int c[4];
#pragma unroll
  for (int dir = 0; dir < 4; ++dir) {
    const int id = env * d.R + dir * d.I + i;
    c[dir] = ring_count(d.leading[id], d.lastcar[id], d.C);
  }
  return (c[0] + c[1] - c[2] - c[3] < 0) ? 1 : 0;
}

// The pull-form advance is exact unless (a) a road pops more than TFX_KP cars, (b) a popped car
// would itself be popped again downstream this tick, or (c) a full ring pops two or more cars
// while receiving pushes (a push could then reuse a popped slot its puller still has to read).
__device__ __forceinline__ bool needs_serial(int kpop, bool any_far, int n_tot, int C) {
  return (kpop > KP) || any_far || (kpop >= 2 && n_tot >= C - 2);
}

// x of the j-th car spawned this tick (j = 0 is xs0): each queues behind the previous one
__device__ __forceinline__ float spawned_x(const Dev &d, float xs0, int j) {
  float xv = xs0;
  for (int q = 0; q < j; ++q) xv = (xv - d.car_l) - d.car_s0;
  return xv;
}

}  // namespace tfx
