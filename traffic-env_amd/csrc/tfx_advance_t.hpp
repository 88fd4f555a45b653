// tfx_advance_t.hpp - the handoff (advance_finished_cars, traffic_env.py:117-135) on the transposed
// layout.  k_move_t already compacted every road (survivors at positions 0 .. m-1) and left the (at
// most TFX_KP = 2) cars that popped in the road's outbox column; what remains per road e is integer
// bookkeeping plus appending the cars its unique predecessor p handed over at positions m, m+1, ...
// The reference's road-order rule is the same as in advance_road (tfx_advance.hpp): p's pushes see
// leading[e] before e's own pops iff p < e.  A road that pops more than two cars, or a handed-off car
// that would itself leave again in the same tick ("far"), sends the env to advance_env_serial_t.
#pragma once
#include "tfx_common.hpp"
#include "tfx_move_t.hpp"

namespace tfx {

__device__ __forceinline__ int advance_road_t(const Dev &d, int env, int e) {
  const int C = d.C;
  const int id = env * d.R + e;
  const int ld = d.leading[id];
  int lc = d.lastcar[id];
  const int4 rc = d.rec[id];
  const int k_e = rec_kpop(rc.x);
  float tail_x = __int_as_float(rc.z);
  const int ld_post = ring_adv(ld, k_e, C);
  int m = rc.w - k_e;  // cars physically on the road after the move (positions 0 .. m-1)

  int ovf = 0;
  const int p = d.pred[e];
  if (p >= 0) {
    const int idp = env * d.R + p;
    const int k_p = rec_kpop(d.rec[idp].x);
    if (k_p > 0) {
      const int ld_seen = (p < e) ? ld : ld_post;
      const size_t pcol = ocol_of(d, env, p), ecol = tcol(d, env, e);
      for (int j = 0; j < k_p; ++j) {  // (k_p <= TFX_KP here: an env with a longer pop run takes the serial form)
        const float2 car = d.outb[pcol + (size_t)j * 64];
        const float xc = car.x - d.length;  // state[e,xi,newlead] -= length (:130)
        const int pos = wrap1(lc + 1, C);
        const float start = (lc != ld_seen) ? (tail_x - d.car_l) - d.car_s0 : INFINITY;
        if (pos != ld_seen) {
          const float xv = (start < xc) ? start : xc;
          d.xv[ecol + (size_t)m * 64] = make_float2(xv, car.y);
          if (d.w) d.w[ecol + (size_t)m * 64] = d.outw[pcol + (size_t)j * 64];
          ++m;
          lc = pos;
          tail_x = xv;
        } else {
          ++ovf;
        }
      }
      d.lastcar[id] = lc;
    }
  }
  if (k_e > 0) d.leading[id] = ld_post;
  d.tailx[id] = tail_x;
  return ovf;
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
__device__ void advance_env_serial_t(const Dev &d, int env, int tick, int tidx) {
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
    float start = INFINITY;
    if (lcn != ldn) {
      float tx, dummy;
      if (phys > 0) tx = d.xv[tpos(d, idn, base + phys - 1)].x;
      else tx = popped(idn, pending - 1, dummy).x;
      start = (tx - d.car_l) - d.car_s0;
    }
    if (pos != ldn) {
      d.xv[tpos(d, idn, base + phys)] = make_float2((start < car.x) ? start : car.x, car.y);
      if (d.w) d.w[tpos(d, idn, base + phys)] = cw;
      d.lastcar[idn] = pos;
    } else {
      if (nr < d.r) rew[nr % d.I] -= d.ovf_pen;
      overflowed = 1;
    }
  };

  // advance_hack :153-154: a car leaving the map records (tick - spawn tick) / 2
  auto trip = [&](float cw) {
    if (d.validate && d.n_trips) {
      const int t = d.n_trips[env];
      if (d.trip_times && t < d.trip_cap) d.trip_times[(size_t)env * d.trip_cap + t] = ((float)tick - cw) / 2.0f;
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
    while (ld != d.lastcar[id] && d.xv[tpos(d, id, 0)].x > d.length) {
      float2 car = d.xv[tpos(d, id, 0)];
      const float cw = d.w ? d.w[tpos(d, id, 0)] : 0.0f;
      const int phys = ring_count(ld, d.lastcar[id], C);
      for (int q = 1; q < phys; ++q) {
        d.xv[tpos(d, id, q - 1)] = d.xv[tpos(d, id, q)];
        if (d.w) d.w[tpos(d, id, q - 1)] = d.w[tpos(d, id, q)];
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
  for (int e = 0; e < d.R; ++e) {
    const int id = env * d.R + e;
    const int n = ring_count(d.leading[id], d.lastcar[id], C);
    d.tailx[id] = (n > 0) ? d.xv[tpos(d, id, n - 1)].x : 0.0f;
  }
  if (overflowed) d.done_tick[env] = tick + 1;
}

}  // namespace tfx
