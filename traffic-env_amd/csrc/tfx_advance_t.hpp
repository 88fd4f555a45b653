// tfx_advance_t.hpp - the handoff (advance_finished_cars, traffic_env.py:117-135) on the transposed
// layout.  k_move_t already compacted every road (survivors at positions 0 .. m-1) and put the cars
// that left into the road's outbox column; what remains per road e is integer bookkeeping plus
// appending the cars its unique predecessor p handed over at positions m, m+1, ...  The reference's
// road-order rule is the same as in advance_road (tfx_advance.hpp): p's pushes see leading[e] before
// e's own pops iff p < e.  Any number of pops per road is handled here; only a handed-off car that
// would itself leave again in the same tick ("far") sends the env to advance_env_serial_t.
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
      const size_t pcol = tcol(d, env, p), ecol = tcol(d, env, e);
      for (int j = 0; j < k_p; ++j) {
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
// travelled more than a road length in one tick).  Follows :117-135 road by road; a road's ring
// content at the time the loop reaches it is: its own popped cars (outbox, in order), then its
// survivors (positions 0..), then whatever earlier roads pushed behind them.
__device__ void advance_env_serial_t(const Dev &d, int env, int tick, int tidx) {
  const int C = d.C;
  int *ob = d.obs + (size_t)env * d.obs_len;
  float *rew = d.rewards + (size_t)env * d.I;
  int overflowed = 0;
  if (!(d.accum_rewards && tidx > 0))
    for (int i = 0; i < d.I; ++i) rew[i] = 0.0f;
  for (int e = 0; e < d.R; ++e) {
    const int sp = d.rec[env * d.R + e].y;
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

  // add_car (:97-114) into road nr; `done_upto` = roads whose own pops have been processed
  auto push = [&](int nr, float2 car, float cw, int done_upto) {
    const int idn = env * d.R + nr;
    const int4 rn = d.rec[idn];
    const int lcn = d.lastcar[idn], ldn = d.leading[idn];
    const int pending = (nr > done_upto) ? rec_kpop(rn.x) : 0;  // popped cars of nr still logically on it
    const int phys = ring_count(ldn, lcn, C) - pending;          // cars physically at positions 0..phys-1
    const int pos = wrap1(lcn + 1, C);
    float start = INFINITY;
    if (lcn != ldn) {
      const float tx = (phys > 0) ? d.xv[tpos(d, idn, phys - 1)].x : d.outb[tpos(d, idn, pending - 1)].x;
      start = (tx - d.car_l) - d.car_s0;
    }
    if (pos != ldn) {
      d.xv[tpos(d, idn, phys)] = make_float2((start < car.x) ? start : car.x, car.y);
      if (d.w) d.w[tpos(d, idn, phys)] = cw;
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
    const int k_e = rec_kpop(d.rec[id].x);
    // the road's own popped cars, in order
    for (int j = 0; j < k_e; ++j) {
      const float cw = d.w ? d.outw[tpos(d, id, j)] : 0.0f;
      if (nr >= 0) {
        ob[e] += 1;
        d.passed_dst[(size_t)env * d.I + e % d.I] = 1;
        float2 car = d.outb[tpos(d, id, j)];
        car.x -= d.length;
        push(nr, car, cw, e - 1);
      } else {
        trip(cw);
      }
      ld = wrap1(ld + 1, C);
      d.leading[id] = ld;
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
