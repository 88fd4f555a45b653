// tfx_advance_t.hpp - the handoff (advance_finished_cars, traffic_env.py:117-135) on the transposed
// layout.  k_move_t already compacted every road (survivors at positions 0 .. m-1) and left the (at
// most TFX_KP = 2) cars that popped in the road's outbox column; what remains per road e is integer
// bookkeeping plus appending the cars its unique predecessor p handed over at positions m, m+1, ...
// The reference's road-order rule is the same as in advance_road (tfx_advance.hpp): p's pushes see
// leading[e] before e's own pops iff p < e.  A road that pops more than two cars, or a handed-off car
// that would itself leave again in the same tick ("far"), sends the env to advance_env_serial_t.
// After the second tick of a two-tick pass (k_edge, tfx_move_tt.hpp) a road that popped keeps its survivors
// where they were: the rows of its column start d.hb[road] rows down, and so do the appends here.
#pragma once
#include "tfx_common.hpp"
#include "tfx_move_t.hpp"

namespace tfx {

// A column whose live rows start `hb` rows down (left so by k_edge) has room for only trows - hb cars: before a
// push would run past the tile's last row, move the n physical cars up to row 0 and clear the offset.
// WP = false: the caller knows the handle has no side-word plane (k_tail<.., W = false>: the rare copy costs it two
// registers, and with them a wavefront per SIMD)
template <bool WP = true>
__device__ __forceinline__ void compact_head_rows(const Dev &d, int id, int hb, int n) {
  for (int q = 0; q < n; ++q) {
    d.xv[tpos(d, id, q)] = d.xv[tpos(d, id, q + hb)];
    if (WP && d.w) d.w[tpos(d, id, q)] = d.w[tpos(d, id, q + hb)];
  }
  d.hb[id] = 0;
}

// Everything advance_road_t reads before it stores anything: the road's own words, its predecessor's pop count and - if
// there are any - the (at most TFX_KP) cars the predecessor left in its outbox column.  Split from the stores so that a
// lane that settles several roads (k_advance: the four approaches of an intersection) can issue all their loads first:
// taken road by road, every road's chain of dependent loads (own words -> pred -> its record -> outbox) waited for the
// stores of the road before it - twelve levels per lane where three do (cfg4 at one env: k_advance 7.3 us per tick).
struct RoadAdvIn {
  int ld, lc, hb, k_p;
  int4 rc;
  float2 car[KP];
  float cw[KP];
};

template <bool HET = false, bool WP = true>
__device__ __forceinline__ RoadAdvIn advance_road_t_load(const Dev &d, int env, int e) {
  RoadAdvIn in;
  const int id = env * d.R + e;
  in.ld = d.leading[id];
  in.lc = d.lastcar[id];
  in.rc = d.rec[id];
  in.hb = 0;
  in.k_p = 0;
#pragma unroll
  for (int j = 0; j < KP; ++j) {
    in.car[j] = make_float2(0.0f, 0.0f);
    in.cw[j] = 0.0f;
  }
  const int p = d.pred[e];
  if (p >= 0) {
    in.k_p = rec_kpop(d.rec[env * d.R + p].x);
    if (in.k_p > 0) {
      in.hb = d.hb[id];
      const size_t pcol = ocol_of(d, env, p);
#pragma unroll
      for (int j = 0; j < KP; ++j)  // (k_p <= TFX_KP here: an env with a longer pop run takes the serial form)
        if (j < in.k_p) {
          in.car[j] = d.outb[pcol + (size_t)j * 64];
          if (d.w) in.cw[j] = d.outw[pcol + (size_t)j * 64];
        }
    }
  }
  return in;
}

template <bool HET = false, bool WP = true>
__device__ __forceinline__ int advance_road_t_apply(const Dev &d, int env, int e, const RoadAdvIn &in) {
  const int C = d.C;
  const int id = env * d.R + e;
  const int ld = in.ld;
  int lc = in.lc;
  const int4 rc = in.rc;
  const int k_e = rec_kpop(rc.x);
  float tail_x = __int_as_float(rc.z);
  const int ld_post = ring_adv(ld, k_e, C);
  int m = rec_ntot(rc.w) - k_e;  // cars physically on the road after the move
  int ta = rec_taila(rc.w);      // heterogeneous cars: table row of the tail (whose length and gap a push queues behind)

  int ovf = 0;
  const int k_p = in.k_p;
  if (k_p > 0) {
    const int p = d.pred[e];
    int hb = in.hb;
    if (hb > 0 && hb + m + k_p > d.trows) {
      compact_head_rows<WP>(d, id, hb, m);
      hb = 0;
    }
    m += hb;  // first free row
    const int ld_seen = (p < e) ? ld : ld_post;
    const size_t ecol = tcol(d, env, e);
#pragma unroll
    for (int j = 0; j < KP; ++j) {
      if (j < k_p) {
        const float2 car = in.car[j];
        const float xc = car.x - d.length;  // state[e,xi,newlead] -= length (:130)
        const int pos = wrap1(lc + 1, C);
        const float tl = HET ? d.arch_tab[ta * ARCH_W + AR_L] : d.car_l, ts0 = HET ? d.arch_tab[ta * ARCH_W + AR_S0] : d.car_s0;
        const float start = (lc != ld_seen) ? (tail_x - tl) - ts0 : INFINITY;
        if (pos != ld_seen) {
          const float xv = (start < xc) ? start : xc;
          d.xv[ecol + (size_t)m * 64] = make_float2(xv, car.y);
          if (d.w) {
            const float cw = in.cw[j];
            d.w[ecol + (size_t)m * 64] = cw;
            if (HET) ta = side_arch(cw);
          }
          ++m;
          lc = pos;
          tail_x = xv;
        } else {
          ++ovf;
        }
      }
    }
    d.lastcar[id] = lc;
  }
  if (k_e > 0) d.leading[id] = ld_post;
  d.tailx[id] = tail_x;
  if (HET) d.taila[id] = ta;
  return ovf;
}

template <bool HET = false, bool WP = true>
__device__ __forceinline__ int advance_road_t(const Dev &d, int env, int e) {
  const RoadAdvIn in = advance_road_t_load<HET, WP>(d, env, e);
  return advance_road_t_apply<HET, WP>(d, env, e, in);
}

// Literal single-thread advance for one env on the transposed layout (an env in which some car
// travelled more than a road length in one tick, or a road popped more than TFX_KP cars).  Follows
// :117-135 road by road.  What a road's column holds when the loop reaches it:
//   compacted road (<= 2 pops):   the popped cars in its outbox column; survivors at rows 0.., then
//                                 whatever earlier roads pushed behind them;
//   uncompacted road (> 2 pops):  the first two popped cars in its outbox column, the others at rows
//                                 2 .. kpop-1, survivors behind them at their old rows, pushes behind
//                                 those; once its pops are processed the column is shifted down by
//                                 kpop, i.e. compacted.
// (inlined on purpose: as a real call it took no SGPR spills but a 512-byte stack frame per lane for the parameter
// block, and k_tail went from 0.19 to 0.43 ms per pair)
template <bool HET = false, bool WP = true>
__device__ __forceinline__ void advance_env_serial_t(const Dev &d, int env, int tick, int tidx) {
  const int C = d.C;
  int *ob = d.obs + (size_t)env * d.obs_len;
  float *rew = d.rewards + (size_t)env * d.I;
  int overflowed = 0;
  if (!(d.accum_rewards && tidx > 0))
    for (int i = 0; i < d.I; ++i) rew[i] = 0.0f;
  for (int e = 0; e < d.R; ++e) {
    const int sp = rec_ovf_sp(d.rec[env * d.R + e].y);
    if (sp > 0) {
      overflowed = 1;
      if (e < d.r)
        for (int j = 0; j < sp; ++j) rew[e % d.I] -= d.ovf_pen;
    }
  }
  if (!(d.agent_mode && tidx > 0))
    for (int e = 0; e < d.r; ++e) ob[e] = 0;
  else
    for (int e = 0; e < d.r; ++e) ob[e] -= rec_kpop(d.rec[env * d.R + e].x);

  // what update_lights and add_car call the tail of a road is tracked in tailx from here on: the last car the
  // move kernel processed (after a two-tick pass the rows themselves are already a tick ahead), then every push
  for (int e = 0; e < d.R; ++e) {
    d.tailx[env * d.R + e] = __int_as_float(d.rec[env * d.R + e].z);
    if (HET) d.taila[env * d.R + e] = rec_taila(d.rec[env * d.R + e].w);
  }
  // row k of a road's column, counted from its first live row
  auto rowb = [&](int idq, int k) { return tpos(d, idq, k + d.hb[idq]); };

  // the j-th popped car of road `idq` while its pops are still pending
  auto popped = [&](int idq, int j, float &cw) {
    if (j >= KP) {  // (only an uncompacted road has that many)
      cw = d.w ? d.w[tpos(d, idq, j)] : 0.0f;
      return d.xv[tpos(d, idq, j)];
    }
    const int envq = idq / d.R;
    const size_t oc = ocol_of(d, envq, idq - envq * d.R) + (size_t)j * 64;
    cw = d.w ? d.outw[oc] : 0.0f;
    return d.outb[oc];
  };

  // add_car (:97-114) into road nr; `done_upto` = roads whose own pops have been processed
  auto push = [&](int nr, float2 car, float cw, int done_upto) {
    const int idn = env * d.R + nr;
    const int4 rn = d.rec[idn];
    const int lcn = d.lastcar[idn], ldn = d.leading[idn];
    const int pending = (nr > done_upto) ? rec_kpop(rn.x) : 0;  // popped cars of nr still logically on it
    const int phys = ring_count(ldn, lcn, C) - pending;          // survivors + cars pushed so far
    const int base = (pending > 0 && rec_unc(rn.y)) ? pending : 0;  // rows the pending cars occupy
    const int pos = wrap1(lcn + 1, C);
    const int ta = HET ? d.taila[idn] : 0;
    const float tl = HET ? d.arch_tab[ta * ARCH_W + AR_L] : d.car_l, ts0 = HET ? d.arch_tab[ta * ARCH_W + AR_S0] : d.car_s0;
    const float start = (lcn != ldn) ? (d.tailx[idn] - tl) - ts0 : INFINITY;
    if (pos != ldn && d.hb[idn] + base + phys >= d.trows) compact_head_rows<WP>(d, idn, d.hb[idn], phys);  // (then base = 0)
    if (pos != ldn) {
      const float xv = (start < car.x) ? start : car.x;
      d.xv[rowb(idn, base + phys)] = make_float2(xv, car.y);
      if (d.w) d.w[rowb(idn, base + phys)] = cw;
      d.lastcar[idn] = pos;
      d.tailx[idn] = xv;
      if (HET) d.taila[idn] = side_arch(cw);
    } else {
      if (nr < d.r) rew[nr % d.I] -= d.ovf_pen;
      overflowed = 1;
    }
  };

  // advance_hack :153-154: a car leaving the map records (tick - spawn tick) / 2
  auto trip = [&](float cw) {
    if (d.validate && d.n_trips) {
      const int t = d.n_trips[env];
      if (d.trip_times && t < d.trip_cap) d.trip_times[(size_t)env * d.trip_cap + t] = side_age(d, tick, cw) / 2.0f;
      d.n_trips[env] = t + 1;
    }
  };

  for (int e = 0; e < d.R; ++e) {
    const int id = env * d.R + e;
    const int nr = d.nexts[e];
    int ld = d.leading[id];
    const int4 re = d.rec[id];
    const int k_e = rec_kpop(re.x);
    // the road's own popped cars, in order
    for (int j = 0; j < k_e; ++j) {
      float cw;
      float2 car = popped(id, j, cw);
      if (nr >= 0) {
        ob[e] += 1;
        d.passed_dst[(size_t)env * d.I + e % d.I] = 1;
        car.x -= d.length;
        push(nr, car, cw, e - 1);
      } else {
        trip(cw);
      }
      ld = wrap1(ld + 1, C);
      d.leading[id] = ld;
    }
    if (k_e > 0 && rec_unc(re.y)) {  // compact: survivors and pushed cars move down by k_e rows
      const int phys = ring_count(ld, d.lastcar[id], C);
      for (int q = 0; q < phys; ++q) {
        d.xv[tpos(d, id, q)] = d.xv[tpos(d, id, q + k_e)];
        if (d.w) d.w[tpos(d, id, q)] = d.w[tpos(d, id, q + k_e)];
      }
    }
    // cars pushed onto an (otherwise emptied) road that are themselves beyond its end
    // (a road with survivors never enters: its head stayed because x <= length, in either tick of a pass)
    while (ld != d.lastcar[id] && d.xv[rowb(id, 0)].x > d.length) {
      float2 car = d.xv[rowb(id, 0)];
      const float cw = d.w ? d.w[rowb(id, 0)] : 0.0f;
      const int phys = ring_count(ld, d.lastcar[id], C);
      for (int q = 1; q < phys; ++q) {
        d.xv[rowb(id, q - 1)] = d.xv[rowb(id, q)];
        if (d.w) d.w[rowb(id, q - 1)] = d.w[rowb(id, q)];
      }
      ld = wrap1(ld + 1, C);
      d.leading[id] = ld;
      if (nr >= 0) {
        ob[e] += 1;
        d.passed_dst[(size_t)env * d.I + e % d.I] = 1;
        car.x -= d.length;
        push(nr, car, cw, e);
      } else {
        trip(cw);
      }
    }
  }
  if (overflowed) d.done_tick[env] = tick + 1;
}

}  // namespace tfx
