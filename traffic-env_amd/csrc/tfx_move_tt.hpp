// tfx_move_tt.hpp - k_move_tt + k_edge: TWO ticks per pass over the cars (transposed layout).
//
// k_move_t reads and writes every live car once per tick and is bound by that traffic; its
// vector ALU sits idle about a third of the time.  Under the Jacobi update a car's next state needs only
// its own and its leader's current state, so while the walk of tick t holds the new states y(k-2), y(k-1)
// of two consecutive cars in registers it can already take car k-1 through tick t+1 - z(k-1) - and store
// THAT: two ticks of arithmetic for one trip of the cars through HBM.  What tick t+1 needs from OTHER
// roads touches only the two ends of a column:
//   * the head car follows the fake leader, whose x comes from the light state and the tail of the next
//     road after tick t's handoff (update_lights :81-94);
//   * cars handed over by tick t's advance, and cars spawned at tick t+1, queue behind the tail.
// Those few cars ("deferred": the head and whatever joined behind the survivors) are left at their tick-t
// state by the pass and taken through tick t+1 by k_edge, a lane-per-road kernel that runs after tick t's
// k_advance: lights and spawns of t+1 (prep_road, unchanged), the deferred cars' IDM steps, the counts,
// the pops of t+1 and the road record k_advance(t+1) consumes.  Launch sequence of a pair:
//   k_move_tt<true>(t)  k_advance(t)  [inputs of t+1]  k_edge(t+1)  k_advance(t+1)
// and the state after it is bit for bit the state after k_move_t, k_advance, k_move_t, k_advance - except
// that a road that popped in t+1 is not compacted: its live rows start d.hb[road] rows down (shifting a column
// is uncoalesced work; reading past two rows is free).  The next move kernel - another pair, or k_move_tt<false>,
// the one-tick form, which every handle big enough for the pairs uses for ALL its single ticks - reads from there
// and writes the column compacted; k_advance, the serial advance, tfx_export_ring and k_refresh honour the offset,
// tfx_import_ring and the resets clear it.
//
// Exactness notes: the interior cars' second step uses idm_step_fast under the same wave-wide domain test
// as the first (both bit-identical to idm_step); k_edge uses idm_step.  The wrapped-ring quirk of the
// waiting count (:210: x, not v, is tested on ring slots 1..lastcar) depends only on the head slot: car i
// behind the head sits in slot leading+1+i, wrapped iff i >= C-1-leading, and an unwrapped ring has no
// such car - so the pass can count tick t+1 before lastcar(t+1) is known.
//
// Agent steps (tfx_agent_step: an env that overflows stands still for the rest of the step) use the pairs too.
// The pass has already moved the interior cars of tick t+1 when tick t's advance discovers an overflow, so envs
// in which tick t COULD overflow are sorted out beforehand: k_risk marks an env "risky" when some ring could run
// full this tick - its cars + its arrivals + the cars its predecessor can hand over exceed the ring - or when a
// road could pop more than TFX_KP cars or hand over a car that runs through the whole next road (then the number
// of pushes is not bounded by TFX_KP).  "Can pop" is a bound, not the IDM result: a car moves at most
// rate*v + a*rate^2/2 per tick because the acceleration never exceeds a.  Risky envs take both ticks of the pair
// one at a time (the pass treats their tiles as the one-tick form, k_edge skips them, a one-tick launch
// restricted to them follows); every other env provably has no overflow in tick t.  (An env that overflows in the
// second tick simply stands still with its row offsets, like any other column between two launches.)
#pragma once
#include <type_traits>

#include "tfx_common.hpp"
#include "tfx_move_t.hpp"

namespace tfx {

// AGENT: inside an agent step - frozen envs are skipped, `passed` accumulates over the step, tiles of risky
// envs (env_risk == tick + 1) take the one-tick form; only_risky: nothing but the tiles of envs marked risky
// for the pair that began one tick earlier (the second tick of those envs, where k_edge finishes the pairs: behind
// k_tail that tick runs inside k_tail itself)
#ifndef TT_P
#define TT_P 4
#endif
// Six wavefronts per SIMD (<= 80 vector registers).  Measured at cfg2 (ms per pass, alone on the chip / vehicle-updates
// per second of the split call): 5 per SIMD (81 registers, what the compiler picks) 0.764 / 4.92e11, 6: 0.745 / 5.22e11,
// 7: 0.728 / 5.04e11, 8: 0.746 / 4.98e11 - at 6 the pass leaves room for one k_tail wavefront per SIMD beside it.
#ifndef TT_WAVES
#define TT_WAVES 6
#endif
// With the side-word plane (validate mode, heterogeneous cars): five per SIMD (<= 96 registers; the kernels would take
// 105-128).  Measured at cfg2, ms per tick of the split call at 4 / 5 / 6 per SIMD: validate mode 0.637 / 0.602 / 0.597,
// three archetypes 0.953 / 0.900-0.908 / 0.975 (the arithmetic of the heterogeneous step pays for the spills of 6).
#ifndef TT_WAVES_W
#define TT_WAVES_W 5
#endif
#define TT_ATTR(W) __attribute__((amdgpu_waves_per_eu((W) ? TT_WAVES_W : TT_WAVES, (W) ? TT_WAVES_W : TT_WAVES)))
// W: the spawn-tick plane travels with the cars (validate mode, advance_hack's trip times :139-157): a car's side word
// is stored wherever its (x, v) is
// HET (implies W): heterogeneous cars, as in k_move_t - the side word is 8 * spawn tick + the car's row of the archetype
// table (LDS copy); a car's own row gives its IDM parameters in BOTH of its ticks, its leader's row the length the gap
// subtracts (the walk carries the lengths of cars k-1 and k-2 along with their new states)
// One tile (64 roads of one env, lane = road) of a pass: every car through tick `tick` and - `two` - all but the roads'
// heads and joiners through tick + 1 as well.  Returns the lane's vehicle-updates of tick `tick`.  Reads and writes the
// ring words (leading, lastcar, tailx, rec, the light words) through `d`.
// CREC: k_tail follows this (two-tick) pass - the road record goes out in its 8-byte form (Dev::crec)
// RSW (with CREC, plain cars, outside agent steps): not the first pair of its call - the ring indices and the row offset
// come from the road state words the k_tail before it left (Dev::rsw)
template <bool TWO, bool AGENT, bool W, bool HET, bool CREC = false, bool RSW = false>
__device__ __forceinline__ int move_tt_tile(const Dev &d, const long tile, const int env, const int lane, const int tick,
                                            const int tick_sp, const int tidx, const bool two, const float *s_arch,
                                            const bool skip = false) {
  constexpr int P = TT_P;
  const int C = d.C;
  const int e_slot = d.slot_road[(int)(tile - (long)env * d.G) * 64 + lane];
  const bool valid = e_slot >= 0;
  const int e = valid ? e_slot : 0;
  const int id = env * d.R + e;
  // (no early exit for a tile that is skipped or stands still: the words that say so are loaded side by side with the
  // rest of the prologue instead of ahead of it - an early `continue` cost the agent pass 9 % of its time)
  const int hb0 = (valid && !RSW) ? d.hb[id] : 0;  // rows the second tick of the last pair left empty at the top of the column
  const bool run = valid && !skip && !(AGENT && env_frozen(d, env, tick));
  const RoadPrep p = prep_road<RSW>(d, id, env, e, tick, tick_sp, tidx, run, run);
  const int hb = run ? (RSW ? p.hb : hb0) : 0;
  if (!RSW && hb) d.hb[id] = 0;       // (this walk writes the column compacted; nobody else reads the byte meanwhile)
  const int n_old = run ? p.n_old : 0;
  const int n_sp = run ? p.n_tot - p.n_old : 0;

  float2 *col = d.xv + ((size_t)tile * d.trows) * 64 + lane;  // written: row k of this road = col[k * 64]
  const float2 *colr = col + (size_t)hb * 64;                 // read: the live rows start hb rows down
  float2 *ocol = d.outb + ((size_t)tile * KP) * 64 + lane;
  float *wcol = W ? d.w + ((size_t)tile * d.trows) * 64 + lane : nullptr;  // side words: same rows as col
  float *owcol = W ? d.outw + ((size_t)tile * KP) * 64 + lane : nullptr;

  int kmax = n_old;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int o = __shfl_xor(kmax, off, 64);
    kmax = o > kmax ? o : kmax;
  }
  kmax = __builtin_amdgcn_readfirstlane(kmax);

  float xprev = p.xL, vprev = 0.0f, llv = 0.0f;  // OLD state of the car ahead (Jacobi); starts as the fake leader
  float y1x = 0.0f, y1v = 0.0f, y2x = 0.0f, y2v = 0.0f;  // NEW states of cars k-1 and k-2
  float y1w = 0.0f;                                      // side word of car k-1
  float y1l = 0.0f, y2l = 0.0f;                          // HET: lengths of cars k-1 and k-2
  int last_a = 0;                                        // HET: table row of the last car processed (the road's tail)
  int kpop = 0, n_wait = 0, n_det = 0, n_wait1 = 0, n_det1 = 0;
  bool open = true, far = false;
  bool pend = false, pend_int = false;  // car k-1 survived tick t and is still to be stored; it is not the new head
  float2 *wp = col;  // where the next surviving car goes: one row further down per survivor
  int kq1 = 0x7fffffff;  // tick t+1 tests x instead of v from this car on (index of tick t)
  float tail_x = 0.0f, tail_z = 0.0f;
  const int kq = (p.ld > p.lc) ? C - 1 - p.ld : 0x7fffffff;

  auto ld2 = [&](const float2 *ptr) {
    const f2v t = __builtin_nontemporal_load(reinterpret_cast<const f2v *>(ptr));
    return make_float2(t.x, t.y);
  };
  auto st2 = [&](float2 *ptr, float a, float b) {
    f2v t;
    t.x = a;
    t.y = b;
    __builtin_nontemporal_store(t, reinterpret_cast<f2v *>(ptr));
  };
  // Car k of this lane's road through tick t, car k-1 through tick t+1.  `last` (a compile-time flag): only
  // the second half, for the road's last car.
  auto step = [&](int k, float x, float v, float sw, auto last) {
    constexpr bool LAST = decltype(last)::value;
    float xn = 0.0f, vn = 0.0f, zx = 0.0f, zv = 0.0f;
    // (one wave-wide test per row.  Round 3 measured the alternatives on this kernel: the speeds read from memory
    // tested once per group of P rows, 0.772 ms per pass against 0.764; no test at all - results wrong - 0.734)
    const bool bad = !HET && ((!LAST && !idm_fast_domain(v)) || (TWO && two && !idm_fast_domain(y1v)));
    const bool off_domain = !HET && __builtin_amdgcn_ballot_w64(bad) != 0ull;
    float my_l = d.car_l;
    if (HET) {
      if (!LAST) {
        last_a = side_arch(sw);
        const float *me = s_arch + last_a * ARCH_W;
        idm_step_het(d, me, x, v, xprev, vprev, llv, xn, vn);
        my_l = me[AR_L];
      }
      if (TWO) idm_step_het(d, s_arch + side_arch(y1w) * ARCH_W, y1x, y1v, y2x, y2v, y2l, zx, zv);
    } else if (d.fastdiv && !off_domain) {
      if (!LAST) idm_step_fast(d, x, v, xprev, vprev, llv, xn, vn);
      if (TWO) idm_step_fast(d, y1x, y1v, y2x, y2v, d.car_l, zx, zv);
    } else {
      if (!LAST) idm_step(d, x, v, xprev, vprev, llv, xn, vn);
      if (TWO) idm_step(d, y1x, y1v, y2x, y2v, d.car_l, zx, zv);
    }
    if (TWO && two && pend) {  // car k-1: the new head keeps its tick-t state (k_edge moves it), the others are a tick ahead
      st2(wp, pend_int ? zx : y1x, pend_int ? zv : y1v);
      if (W) wcol[wp - col] = y1w;
      wp += 64;
      if (pend_int) {
        const float wq1 = (k - 1 >= kq1) ? zx : zv;
        n_wait1 += (wq1 < d.thresh) ? 1 : 0;
        n_det1 += (zx > d.near_end) ? 1 : 0;
        if (LAST) tail_z = zx;  // (the road's last car is flushed by the LAST call)
      }
    }
    if (LAST) return;
    xprev = x;
    vprev = v;
    llv = my_l;
    const bool was_open = open;
    const bool pop = open && (xn > d.length);  // the while loop of :123
    open = pop;
    if (pop) {
      if (kpop < KP) {
        ocol[(size_t)kpop * 64] = make_float2(xn, vn);
        if (W) owcol[(size_t)kpop * 64] = sw;
      } else {  // third pop: no survivor has been written yet - from here on every car stays in its row
        st2(&col[(size_t)k * 64], xn, vn);
        if (W) wcol[(size_t)k * 64] = sw;
        wp = col + (size_t)(k + 1) * 64;
      }
      far = far || ((xn - d.length) > d.length);
      ++kpop;
    } else if (TWO && two) {
      pend = true;
      pend_int = !was_open;
      if (was_open) kq1 = C - 1 - ring_adv(p.ld, kpop, C) + kpop;  // (kpop is final: this is the first survivor)
    } else {
      st2(wp, xn, vn);
      if (W) wcol[wp - col] = sw;
      wp += 64;
    }
    const float wq = (k >= kq) ? xn : vn;
    n_wait += (wq < d.thresh) ? 1 : 0;
    n_det += (xn > d.near_end) ? 1 : 0;
    if (!TWO) tail_x = xn;  // (a pair: y1 holds the last car's new state when the walk ends)
    y2x = y1x;
    y2v = y1v;
    y1x = xn;
    y1v = vn;
    if (W) y1w = sw;
    if (HET) {
      y2l = y1l;
      y1l = my_l;
    }
  };

  // ---- cars in memory: rows 0 .. kmax-1 of the live part, P rows in flight --------------------
  float2 pf[P];
  float pfw[P];
  const float *wcolr = wcol + (size_t)hb * 64;
#pragma unroll
  for (int u = 0; u < P; ++u) {
    pf[u] = (u < n_old) ? ld2(&colr[(size_t)u * 64]) : make_float2(0.0f, 0.0f);
    pfw[u] = (W && u < n_old) ? wcolr[(size_t)u * 64] : 0.0f;
  }
  for (int k0 = 0; k0 < kmax; k0 += P) {
#pragma unroll
    for (int u = 0; u < P; ++u) {
      const int k = k0 + u;
      if (k < kmax) {
        const float2 cur = pf[u];
        const float curw = pfw[u];
        if (k + P < kmax) {
          pf[u] = (k + P < n_old) ? ld2(&colr[(size_t)(k + P) * 64]) : make_float2(0.0f, 0.0f);
          pfw[u] = (W && k + P < n_old) ? wcolr[(size_t)(k + P) * 64] : 0.0f;
        }
        if (k < n_old) step(k, cur.x, cur.y, curw, std::false_type{});
      }
    }
  }
  // ---- cars spawned this tick (add_car :97-114): they queue behind the tail ------------------
  if (__builtin_amdgcn_ballot_w64(n_sp > 0) != 0ull) {
    int smax = n_sp;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int o = __shfl_xor(smax, off, 64);
      smax = o > smax ? o : smax;
    }
    smax = __builtin_amdgcn_readfirstlane(smax);
    if (HET) {
      // add_car :97-114 car by car: each queues behind the road's tail at the moment it is created - the tail's
      // OWN length and minimum gap - and brings the row add_new_cars drew for it (:164); as in k_move_t
      int lcq = ring_adv(p.ld, n_old, C);
      float tx = d.tailx[id];
      int ta = d.taila[id];
      const int ej = d.entry_idx[e];
      const uint8_t *rows = (d.spawn_arch && d.spawn_mode == TFX_SPAWN_COUNTS && ej >= 0)
                                ? d.spawn_arch + (size_t)tidx * d.spawn_arch_stride +
                                      ((size_t)env * d.n_entry + ej) * d.spawn_arch_S
                                : nullptr;
      for (int s = 0; s < smax; ++s)
        if (s < n_sp) {
          const int row = (rows && s < d.spawn_arch_S) ? (rows[s] & (TFX_MAX_ARCH - 1)) : 0;
          const float start = (lcq != p.ld) ? (tx - s_arch[ta * ARCH_W + AR_L]) - s_arch[ta * ARCH_W + AR_S0] : INFINITY;
          const float xs = (start < 0.0f) ? start : 0.0f;
          step(n_old + s, xs, s_arch[row * ARCH_W + AR_V], side_pack(tick, row), std::false_type{});
          lcq = wrap1(lcq + 1, C);
          tx = xs;
          ta = row;
        }
    } else {
      for (int s = 0; s < smax; ++s)
        if (s < n_sp) step(n_old + s, spawned_x(d, p.xs0, s), d.car_v, (float)tick, std::false_type{});
    }
  }
  // ---- the road's last car through tick t+1 ----------------------------------------------------
  if (TWO && two && __builtin_amdgcn_ballot_w64(pend) != 0ull) {
    if (pend) step(p.n_tot, 0.0f, 0.0f, 0.0f, std::true_type{});
  }

  // ---- phase W -------------------------------------------------------------------------------
  if (run) {
    const int n_tot = p.n_tot;
    if (e < d.r) {
      int *ob = d.obs + (size_t)env * d.obs_len;
      if (n_tot > 0 && !two) {  // (a pair: edge_tile adds both ticks' waiting counts at once and stores `detected` -
        d.waiting[(size_t)env * d.r + e] += n_wait;  //  this tick's only if the road is empty in the next, :199-201)
        ob[d.r + e] = n_det;
      }
      // accumulates over the agent step (a pair: edge_tile adds both ticks' pops at once)
      if (AGENT && !two) ob[e] = (tidx > 0) ? ob[e] + kpop : kpop;
      else if (!two) ob[e] = kpop;  // (a pair: overwritten by the second tick before anyone reads it)
      if (kpop > 0) d.passed_dst[(size_t)env * d.I + e % d.I] = 1;
    }
    if (TWO) tail_x = y1x;  // new x of the last car processed (0 if there was none)
    if (TWO && CREC) {  // (k_tail follows: 8 bytes instead of 16)
      d.crec[id] = make_int2(crec_pack<HET>(kpop, n_tot, p.ld, p.lc, HET ? last_a : 0, kpop > KP, p.ovf_sp > 0), __float_as_int(tail_x));
      if (p.ovf_sp > 0) d.ovf_cnt[id] = p.ovf_sp;
    } else {
      d.rec[id] = make_int4(rec_pack(kpop, p.ld, C), rec_y(p.ovf_sp, kpop > KP), __float_as_int(tail_x),
                            n_tot | (HET ? last_a << 16 : 0));
    }
    if (two) {
      d.rec2f[id] = make_float2(y1v, tail_z);
      d.rec2c[id] = rec2c_pack(n_wait + n_wait1, n_det1, n_det, n_tot > 0);
    }
    if (far || kpop > KP) d.env_flag[env] = tick + 1;
    if (!two) d.leadx[id] = p.xL;  // (read by tfx_export_ring only: the second tick of a pair writes its own)
    return n_tot;
  }
  return 0;
}

template <bool TWO, bool AGENT = false, bool W = false, bool HET = false, bool CREC = false, bool RSW = false>
__global__ __launch_bounds__(256) TT_ATTR(W) void k_move_tt(const Dev d, const int tidx, const int only_risky) {
  static_assert(!CREC || TWO, "only a two-tick pass is followed by k_tail");
  static_assert(!RSW || (CREC && !AGENT && !HET), "road state words: between the pairs of a plain tfx_step call");
  static_assert(!HET || W, "heterogeneous cars carry their table row in the side word");
  __shared__ float s_arch[HET ? TFX_MAX_ARCH * ARCH_W : 1];
  if (HET) load_arch(d, s_arch);
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tick = *d.tickA;
  const long tiles = (long)d.E * d.G;
  const long nw = (long)gridDim.x * 4;
  const int tick_sp = (d.spawn_mode == TFX_SPAWN_PERIODIC) ? tick % d.spawn_period : 0;

  if (AGENT && only_risky && risk_any_word(d, tidx) != tick) return;  // (k_edge / k_tail has moved every other env and the clock)

  unsigned long long my_updates = 0;

  for (long tile = (long)blockIdx.x * 4 + wv; tile < tiles; tile += nw) {
    const int env = (int)(tile / d.G);  // a tile never straddles envs
    if (AGENT && only_risky && risk_word(d, env, tidx) != tick) continue;
    // the envs k_risk sorted out of this pair (their first tick could overflow).  Where k_tail follows, it takes them
    // through both ticks itself, one at a time, and the pass leaves their tiles alone - every tile it does walk goes
    // through two ticks, known at compile time.  Otherwise their tiles take the one-tick form here (wave-uniform) and
    // a restricted launch brings the second tick.
    const bool sorted_out = AGENT && TWO && risk_word(d, env, tidx) == tick + 1;
    const bool two = TWO && (CREC || !sorted_out);
    // (a stamp left by an earlier run at the same tick number - the clock can be set back - is honoured all the
    // way: this pass takes the tile one tick at a time, k_edge skips it, and the restricted launch must come)
    if (AGENT && TWO && !two && lane == 0) risk_any_word(d, tidx) = tick + 1;
    my_updates += (unsigned long long)move_tt_tile<TWO, AGENT, W, HET, CREC, RSW>(d, tile, env, lane, tick, tick_sp, tidx, two, s_arch,
                                                                            CREC && sorted_out);
  }

  for (int off = 32; off > 0; off >>= 1) my_updates += __shfl_down(my_updates, off);
  if (lane == 0 && my_updates) veh_add(d.veh, my_updates);
  if (blockIdx.x == 0 && threadIdx.x == 0) *d.tickB = tick;
}

// Could the tick starting at `tick` overflow a ring of this lane's road - or pop / hand over more cars than a pair's
// bookkeeping assumes (see the head of this file)?  A bound, not the IDM result: a car moves at most
// rate * v + a rate^2 / 2 per tick because the IDM acceleration never exceeds a (:56-57: a * (1 - q^delta - u^2) with
// q >= 0; heterogeneous cars: the table's largest a).  Reads the ring words through `d` (k_tail: its LDS copies) and the
// road's first cars from the rows as they stand (after a pair: already at `tick`).
__device__ __forceinline__ bool risk_lane(const Dev &d, long tile, int env, int lane, int tick, int tick_sp, int tidx) {
  const int C = d.C;
  const int e = d.slot_road[(int)(tile - (long)env * d.G) * 64 + lane];
  if (e < 0) return false;
  const float half_ar2 = (0.5f * (d.risk_a * d.rate)) * d.rate;
  const int id = env * d.R + e;
  const int n = ring_count(d.leading[id], d.lastcar[id], C);
  const int ej = d.entry_idx[e];
  const int c_sp = ej >= 0 ? spawn_count(d, env, e, ej, tick_sp, tidx) : 0;
  bool risky = n + c_sp > C - 2;
  // how many cars could leave: a prefix of the cars that can reach the end of the road at all
  const float2 *col = d.xv + ((size_t)tile * d.trows) * 64 + lane + (size_t)d.hb[id] * 64;
  int pops = 0;
  const int n_look = (d.riskhint && d.riskhint[id] == 0) ? 0 : n;  // (0: edge_tile saw that the head cannot leave)
  for (int j = 0; j <= KP && j < n_look; ++j) {
    const float2 c = col[(size_t)j * 64];
    const float reach = c.x + __builtin_fmaxf(d.rate * c.y + half_ar2, 0.0f);
    if (!(reach > d.length)) break;
    ++pops;
    if ((reach - d.length) > d.length) risky = true;  // could run through the next road as well
  }
  if (pops > KP) risky = true;
  const int nx = d.nexts[e];
  if (pops > 0 && nx >= 0) {
    const int idn = env * d.R + nx;
    if (ring_count(d.leading[idn], d.lastcar[idn], C) + pops > C - 2) risky = true;
  }
  return risky;
}

// Before a pair inside an agent step: marks the envs in which the pair's first tick could overflow (risk_lane).  Only
// the FIRST pair of a decision needs this launch: k_tail<AGENT> evaluates the same bound for the pair that follows it
// as its last phase (tfx_tail.hpp).
__global__ __launch_bounds__(256) void k_risk(const Dev d, const int tidx) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tick = *d.tickA;
  const long tiles = (long)d.E * d.G;
  const long nw = (long)gridDim.x * 4;
  const int tick_sp = (d.spawn_mode == TFX_SPAWN_PERIODIC) ? tick % d.spawn_period : 0;
  for (long tile = (long)blockIdx.x * 4 + wv; tile < tiles; tile += nw) {
    const int env = (int)(tile / d.G);
    if (env_frozen(d, env, tick)) continue;
    if (risk_lane(d, tile, env, lane, tick, tick_sp, tidx)) {
      risk_word(d, env, tidx) = tick + 1;
      risk_any_word(d, tidx) = tick + 1;
    }
  }
}

// The second tick of a pair for the cars k_move_tt<true> could not take through it (see the head of this
// file), for the 64 roads of one tile: lane = road.  Runs after the advance of the first tick; same tile / lane
// ownership as the pass, a few cars per road.  Returns the lane's vehicle-updates.
// HET: heterogeneous cars - `arch` is the caller's LDS copy of the archetype table; a deferred car's row comes from its
// side word (or from the spawner's draw), its leader's length travels along with the leader's old state
// full_out = false (k_tail, every pair of a tfx_step call but the last): what nobody can read before the next pair
// overwrites it is not stored - `passed` of the tick (outside agent steps, where it accumulates) and the fake leader's x
// kept for tfx_export_ring; `detected` is stored all the same (an empty road keeps its last value, :199-201)
template <bool AGENT, bool W = false, bool HET = false>
__device__ __forceinline__ int edge_tile(const Dev &d, long tile, int env, int lane, int tick, int tick_sp, int tidx,
                                         const float *arch = nullptr, const bool full_out = true) {
  const int C = d.C;
  // (agent step: frozen envs stand still; risky envs took the first tick alone and get the second from a
  // one-tick launch of their own)
  if (AGENT && (risk_word(d, env, tidx) == tick || env_frozen(d, env, tick))) return 0;
  const int e = d.slot_road[(int)(tile - (long)env * d.G) * 64 + lane];
  if (e < 0) return 0;
  const int id = env * d.R + e;
  const int4 rc = d.rec[id];    // the pass's record of the first tick
  const int r2c = d.rec2c[id];   // the road's waiting count of the first tick plus the second tick's so far | detected so far << 16
  const float2 r2f = d.rec2f[id];  // the tail's v after the first tick, its x after the second
  const RoadPrep p = prep_road(d, id, env, e, tick, tick_sp, tidx, true, true);
  const int m0 = rec_ntot(rc.w) - rec_kpop(rc.x);  // survivors of the first tick: rows 0 .. m0-1 (row 0 = the head)
  const int n_old = p.n_old, n_tot = p.n_tot;

  float2 *col = d.xv + ((size_t)tile * d.trows) * 64 + lane;
  float2 *ocol = d.outb + ((size_t)tile * KP) * 64 + lane;
  float *wcol = W ? d.w + ((size_t)tile * d.trows) * 64 + lane : nullptr;
  float *owcol = W ? d.outw + ((size_t)tile * KP) * 64 + lane : nullptr;

  int kpop = 0, n_wait = r2c & 1023, n_det = (r2c >> 10) & 511;
  bool open = true, far = false;
  const int kq = C - 1 - p.ld;
  float lx = p.xL, lv = 0.0f, ll = 0.0f;  // OLD state of the car ahead of the next deferred car
  float tail_x = 0.0f;
  float hx = 0.0f, hv = 0.0f;  // the new state of the road's head (the first deferred car)
  int last_a = HET ? rec_taila(rc.w) : 0;  // HET: table row of the road's last car so far
  // deferred car i (its tick-t state x, v) through tick t+1 against (lx, lv, ll); fresh: spawned this tick (side word sw)
  auto car = [&](int i, float x, float v, bool fresh, float sw = 0.0f) {
    float zx, zv;
    if (HET) {
      if (!fresh) sw = wcol[(size_t)i * 64];
      last_a = side_arch(sw);
      const float *me = arch + last_a * ARCH_W;
      idm_step_het(d, me, x, v, lx, lv, ll, zx, zv);
      ll = me[AR_L];
    } else {
      if (W && fresh) sw = (float)tick;
      idm_step(d, x, v, lx, lv, ll, zx, zv);
      ll = d.car_l;
    }
    lx = x;
    lv = v;
    if (AGENT && i == 0) {
      hx = zx;
      hv = zv;
    }
    const bool pop = open && (zx > d.length);
    open = pop;
    if (pop && kpop < KP) {
      ocol[(size_t)kpop * 64] = make_float2(zx, zv);
      if (W) owcol[(size_t)kpop * 64] = fresh ? sw : wcol[(size_t)i * 64];
    } else {  // (a popped car beyond the outbox stays in its row: uncompacted)
      col[(size_t)i * 64] = make_float2(zx, zv);
      if (W && fresh) wcol[(size_t)i * 64] = sw;
    }
    if (pop) far = far || ((zx - d.length) > d.length);
    kpop += pop ? 1 : 0;
    const float wq = (i >= kq) ? zx : zv;
    n_wait += (wq < d.thresh) ? 1 : 0;
    n_det += (zx > d.near_end) ? 1 : 0;
    tail_x = zx;
  };
  if (m0 > 0) {
    const float2 head = col[0];
    car(0, head.x, head.y, false);
    // cars behind a head that left: already a tick ahead, the pop prefix may run on into them
    for (int i = 1; i < m0 && open; ++i) {
      const float2 z = col[(size_t)i * 64];
      if (z.x > d.length) {
        if (kpop < KP) {
          ocol[(size_t)kpop * 64] = z;
          if (W) owcol[(size_t)kpop * 64] = wcol[(size_t)i * 64];
        }
        far = far || ((z.x - d.length) > d.length);
        ++kpop;
      } else {
        open = false;
      }
    }
    if (m0 >= 2) {  // whoever queues behind the survivors follows the last one's tick-t state
      lx = __int_as_float(rc.z);
      lv = r2f.x;
      tail_x = r2f.y;
      if (HET) {
        last_a = rec_taila(rc.w);
        ll = arch[last_a * ARCH_W + AR_L];
      }
    }
  }
  for (int i = m0; i < n_old; ++i) {  // handed over by the first tick's advance
    const float2 c = col[(size_t)i * 64];
    car(i, c.x, c.y, false);
  }
  if (HET) {  // spawned this tick, car by car behind the tail's own length and gap (as in the pass)
    if (n_tot > n_old) {
      int lcq = ring_adv(p.ld, n_old, C);
      float tx = d.tailx[id];
      int ta = d.taila[id];
      const int ej = d.entry_idx[e];
      const uint8_t *rows = (d.spawn_arch && d.spawn_mode == TFX_SPAWN_COUNTS && ej >= 0)
                                ? d.spawn_arch + (size_t)tidx * d.spawn_arch_stride +
                                      ((size_t)env * d.n_entry + ej) * d.spawn_arch_S
                                : nullptr;
      for (int s = 0; s < n_tot - n_old; ++s) {
        const int row = (rows && s < d.spawn_arch_S) ? (rows[s] & (TFX_MAX_ARCH - 1)) : 0;
        const float start = (lcq != p.ld) ? (tx - arch[ta * ARCH_W + AR_L]) - arch[ta * ARCH_W + AR_S0] : INFINITY;
        const float xs = (start < 0.0f) ? start : 0.0f;
        car(n_old + s, xs, arch[row * ARCH_W + AR_V], true, side_pack(tick, row));
        lcq = wrap1(lcq + 1, C);
        tx = xs;
        ta = row;
      }
    }
  } else {
    for (int s = 0; s < n_tot - n_old; ++s) car(n_old + s, spawned_x(d, p.xs0, s), d.car_v, true);  // spawned this tick
  }

  const bool unc = kpop > KP;
  if (e < d.r) {
    int *ob = d.obs + (size_t)env * d.obs_len;
    if (n_wait != 0) d.waiting[(size_t)env * d.r + e] += n_wait;
    // `detected` keeps its last value while a road is empty (:199-201): this tick's count, or - the road emptied in this
    // tick - the first tick's, which the pass handed over instead of storing it
    if (n_tot > 0) ob[d.r + e] = n_det;
    else if ((r2c >> 28) & 1) ob[d.r + e] = (r2c >> 19) & 511;
    // (agent step: `passed` accumulates - this tick's pops and the pass's of the tick before; the first pair starts it)
    if (AGENT) ob[e] = (tidx > 1 ? ob[e] : 0) + rec_kpop(rc.x) + kpop;
    else if (full_out) ob[e] = kpop;
    if (kpop > 0) d.passed_dst[(size_t)env * d.I + e % d.I] = 1;
  }
  if (!unc && kpop) d.hb[id] = (uint8_t)kpop;  // (0 before: the pass wrote the column compacted)
  d.rec[id] = make_int4(rec_pack(kpop, p.ld, C), rec_y(p.ovf_sp, unc),
                        __float_as_int(tail_x), n_tot | (HET ? last_a << 16 : 0));
  if (far || unc) d.env_flag[env] = tick + 1;
  if (full_out || AGENT) d.leadx[id] = p.xL;  // (k_tail<AGENT>: its LDS copy - stored when the decision or the env ends)
  if (AGENT && d.riskhint) {
    // can the head leave in the NEXT tick at all (risk_lane's first test, on the state this lane still holds)?  An empty
    // road may receive a car in the advance, a head that left makes the next car the head: those roads load their rows
    const float half_ar2 = (0.5f * (d.risk_a * d.rate)) * d.rate;
    const bool check = n_tot == 0 || kpop > 0 || (hx + __builtin_fmaxf(d.rate * hv + half_ar2, 0.0f)) > d.length;
    d.riskhint[id] = check ? 1 : 0;
  }
  return n_tot;
}

template <bool AGENT, bool W = false, bool HET = false>
__global__ __launch_bounds__(256) void k_edge(const Dev d, const int tidx) {
  __shared__ float s_arch[HET ? TFX_MAX_ARCH * ARCH_W : 1];
  if (HET) load_arch(d, s_arch);
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tick = *d.tickA;
  const long tiles = (long)d.E * d.G;
  const long nw = (long)gridDim.x * 4;
  const int tick_sp = (d.spawn_mode == TFX_SPAWN_PERIODIC) ? tick % d.spawn_period : 0;

  unsigned long long my_updates = 0;
  for (long tile = (long)blockIdx.x * 4 + wv; tile < tiles; tile += nw)
    my_updates += (unsigned long long)edge_tile<AGENT, W, HET>(d, tile, (int)(tile / d.G), lane, tick, tick_sp, tidx, s_arch);

  for (int off = 32; off > 0; off >>= 1) my_updates += __shfl_down(my_updates, off);
  if (lane == 0 && my_updates) veh_add(d.veh, my_updates);
  if (blockIdx.x == 0 && threadIdx.x == 0) *d.tickB = tick;
}

}  // namespace tfx
