// tfx_advance.hpp - k_advance: ring pop + car handoff (advance_finished_cars, traffic_env.py:117-135,
// and its validate-mode twin advance_hack :139-157), plus the per-intersection stores of the tick
// (phase, elapsed, rewards, done).
//
// The reference walks roads sequentially.  Here every road PULLS the cars its unique predecessor
// popped (k_move left them in the per-road record), one lane per intersection (its four incoming
// roads) or exit road.  The pull form is exact whenever every road pops at most TFX_KP cars and no
// handed-off car could be popped again in the same tick; k_move detects the contrary per env and
// that env then goes through advance_env_serial, a literal single-thread restatement - so results
// equal the sequential algorithm in every case.
#pragma once
#include "tfx_common.hpp"
#include "tfx_advance_t.hpp"

namespace tfx {

// Ring pop + handoff for destination road e.  Returns the overflow count of pushes into e.
// What the reference's loop does to road e, given that (a) e's own pops are its first k_e cars and
// (b) the cars pushed into e are the k_p cars its predecessor p popped (read straight from p's ring:
// k_move left their post-move state in slots head_p, head_p+1, ...):
//   - the loop visits roads in ascending order, so p's pushes see leading[e] BEFORE e's own pops
//     when p < e and AFTER them when p > e (both the ring-full test and the empty-road test of
//     add_car :100-105 read leading[e]);
//   - successive pushes queue behind each other: start = x_tail - l - s0 of the previous push.
// Who writes what (no two lanes touch the same word): road e's lane writes e's leading, lastcar,
// tail cache and the slots it pushes into; it also finishes p's pop by copying p's fake-leader x
// into p's new leader slot (:133) - the one slot of p that only e's lane reads.
__device__ __forceinline__ int advance_road(const Dev &d, int env, int e, int tick, int tidx) {
  const int C = d.C;
  const int id = env * d.R + e;
  const int ld = d.leading[id];
  int lc = d.lastcar[id];
  const int4 rc = d.rec[id];
  const int k_e = rec_kpop(rc.x);
  float tail_x = __int_as_float(rc.z);
  const int ld_post = ring_adv(ld, k_e, C);
  float2 *rx = d.xv + (size_t)id * C;

  int ovf = 0;
  const int p = d.pred[e];
  if (p >= 0) {
    const int idp = env * d.R + p;
    const int rpx = d.rec[idp].x;
    const int k_p = rec_kpop(rpx);
    if (k_p > 0) {
      const int ld_seen = (p < e) ? ld : ld_post;
      float2 *px = d.xv + (size_t)idp * C;
      int ps = rec_head(rpx);
      for (int j = 0; j < k_p; ++j) {
        const float2 car = px[ps];
        const float xc = car.x - d.length;  // state[e,xi,newlead] -= length (:130)
        const int pos = wrap1(lc + 1, C);
        const float start = (lc != ld_seen) ? (tail_x - d.car_l) - d.car_s0 : INFINITY;
        if (pos != ld_seen) {
          const float xv = (start < xc) ? start : xc;
          rx[pos] = make_float2(xv, car.y);
          if (d.w) d.w[(size_t)id * C + pos] = d.w[(size_t)idp * C + ps];
          lc = pos;
          tail_x = xv;
        } else {
          ++ovf;
        }
        if (j + 1 < k_p) ps = wrap1(ps + 1, C);
      }
      d.lastcar[id] = lc;
      // p's new leader slot (its last popped car) gets p's fake-leader x (:133): the value k_move
      // used this tick (leadx).  It must NOT be recomputed from obs here - the lane of p's
      // intersection stores the new phase / elapsed in this same kernel, and a lane that reads them
      // afterwards would apply the light update twice (found by the round-1 fuzz run).
      px[ps].x = d.leadx[idp];
    }
  }
  if (k_e > 0) {
    d.leading[id] = ld_post;
    if (e >= d.r) rx[ld_post].x = INFINITY;  // exit roads: nobody pulls, the leader stays at +inf
  }
  d.tailx[id] = tail_x;
  return ovf;
}

// Literal single-thread advance for one env (taken when a road popped more than TFX_KP cars or a
// handed-off car could itself be popped again this tick).  Follows :117-157 line by line.
__device__ void advance_env_serial(const Dev &d, int env, int tick, int tidx) {
  const int C = d.C;
  int *ob = d.obs + (size_t)env * d.obs_len;
  float *rew = d.rewards + (size_t)env * d.I;
  int overflowed = 0;
  if (!(d.accum_rewards && tidx > 0))
    for (int i = 0; i < d.I; ++i) rew[i] = 0.0f;
  for (int e = 0; e < d.R; ++e) {
    const int sp = rec_ovf_sp(d.rec[env * d.R + e].y);  // spawn overflows happened before move_cars
    if (sp > 0) {
      overflowed = 1;
      if (e < d.r)
        for (int j = 0; j < sp; ++j) rew[e % d.I] -= d.ovf_pen;
    }
  }
  if (!(d.agent_mode && tidx > 0))
    for (int e = 0; e < d.r; ++e) ob[e] = 0;
  else  // k_move already added this tick's pops of the parallel form: take them back out
    for (int e = 0; e < d.r; ++e) ob[e] -= rec_kpop(d.rec[env * d.R + e].x);
  for (int e = 0; e < d.R; ++e) {
    const int id = env * d.R + e;
    float2 *rx = d.xv + (size_t)id * C;
    float *rw = d.w ? d.w + (size_t)id * C : nullptr;
    int ld = d.leading[id];
    while (ld != d.lastcar[id] && rx[wrap1(ld + 1, C)].x > d.length) {
      const int newlead = wrap1(ld + 1, C);
      const int nr = d.nexts[e];
      if (nr >= 0) {
        ob[e] += 1;
        d.passed_dst[(size_t)env * d.I + e % d.I] = 1;
        rx[newlead].x -= d.length;
        const int idn = env * d.R + nr;
        float2 *nx = d.xv + (size_t)idn * C;
        const int lcn = d.lastcar[idn], ldn = d.leading[idn];
        const int pos = wrap1(lcn + 1, C);
        const float start = (lcn != ldn) ? (nx[lcn].x - d.car_l) - d.car_s0 : INFINITY;
        if (pos != ldn) {
          const float xc = rx[newlead].x;
          nx[pos] = make_float2((start < xc) ? start : xc, rx[newlead].y);
          if (rw) d.w[(size_t)idn * C + pos] = rw[newlead];
          d.lastcar[idn] = pos;
        } else {
          if (nr < d.r) rew[nr % d.I] -= d.ovf_pen;
          overflowed = 1;
        }
      } else if (d.validate && d.n_trips) {
        const int t = d.n_trips[env];
        if (d.trip_times && t < d.trip_cap)
          d.trip_times[(size_t)env * d.trip_cap + t] = ((float)tick - (rw ? rw[newlead] : 0.0f)) / 2.0f;
        d.n_trips[env] = t + 1;
      }
      rx[newlead].x = rx[ld].x;
      ld = newlead;
      d.leading[id] = ld;
    }
  }
  for (int e = 0; e < d.R; ++e) {
    const int id = env * d.R + e;
    d.tailx[id] = d.xv[(size_t)id * C + d.lastcar[id]].x;
  }
  if (overflowed) d.done_tick[env] = tick + 1;
}

// One work item of the advance: s < I is intersection s of the env (its four incoming roads s, I+s, 2I+s, 3I+s,
// roadgraph.py:38-39, plus the light words and the reward of that intersection), s >= I is exit road r + (s - I).
// Item 0 of an env flagged for it runs the literal serial loop for the whole env.
// WP = false: the caller knows there is no side-word plane (compact_head_rows)
// BATCH (transposed layout): the loads of the intersection's four roads are issued before any of their stores
// (advance_road_t_load; k_advance - k_tail works on LDS copies and within the registers of a wavefront of the pass)
template <bool TL, bool HET = false, bool GREEDY = false, bool WP = true, bool BATCH = false>
__device__ __forceinline__ void advance_item(const Dev &d, int env, int s, int tick, int tidx) {
  const bool frozen = env_frozen(d, env, tick);  // stopped for the rest of this agent step
  const bool serial = !frozen && d.env_flag[env] == tick + 1;
  // The greedy controller's decision for the NEXT tick, from the counts this tick leaves behind: the lane of
  // intersection s has just settled its four incoming roads (an env on the serial path: its one lane decides for
  // every intersection; an env that stands still keeps deciding from its standing counts, as an agent would).
  const bool decide = GREEDY && d.greedy_spacing > 0 && (tick + 1) % d.greedy_spacing == 0;
  if (frozen) {
    if (decide && s < d.I) d.greedy_act[(size_t)env * d.I + s] = greedy_decide(d, env, s);
    return;
  }
  if (serial && s == 0) {
    if (TL) advance_env_serial_t<HET, WP>(d, env, tick, tidx);
    else advance_env_serial(d, env, tick, tidx);
    if (decide)
      for (int i = 0; i < d.I; ++i) d.greedy_act[(size_t)env * d.I + i] = greedy_decide(d, env, i);
  }
  if (s < d.I) {
    int ph_new, el_new;
    light_update(d, env, s, tick, tidx, ph_new, el_new);
    if (!serial) {
      int ovf = 0;
      if (TL && BATCH) {
        RoadAdvIn in[4];
#pragma unroll
        for (int dir = 0; dir < 4; ++dir) in[dir] = advance_road_t_load<HET, WP>(d, env, dir * d.I + s);
#pragma unroll
        for (int dir = 0; dir < 4; ++dir)
          ovf += advance_road_t_apply<HET, WP>(d, env, dir * d.I + s, in[dir]) + rec_ovf_sp(in[dir].rc.y);
      } else {
#pragma unroll
        for (int dir = 0; dir < 4; ++dir) {
          const int e = dir * d.I + s;
          ovf += (TL ? advance_road_t<HET, WP>(d, env, e) : advance_road(d, env, e, tick, tidx)) + rec_ovf_sp(d.rec[env * d.R + e].y);
        }
      }
      // rewards[:] = 0 (:233) then -= OVERFLOW_PENALTY per dropped car (:110): exact in fp32
      float rw = (d.accum_rewards && tidx > 0) ? d.rewards[(size_t)env * d.I + s] : 0.0f;
      for (int j = 0; j < ovf; ++j) rw -= d.ovf_pen;
      d.rewards[(size_t)env * d.I + s] = rw;
      if (ovf > 0) d.done_tick[env] = tick + 1;
      if (decide) d.greedy_act[(size_t)env * d.I + s] = greedy_decide(d, env, s);
    }
    int *ob = d.lights + (size_t)env * d.lights_stride;
    ob[s] = ph_new;
    ob[d.I + s] = el_new;
  } else if (!serial) {
    const int e = d.r + (s - d.I);
    const int ovf = TL ? advance_road_t<HET, WP>(d, env, e) : advance_road(d, env, e, tick, tidx);
    if (ovf > 0) d.done_tick[env] = tick + 1;
    if (d.validate && d.n_trips && s == d.I) {
      // advance_hack :153-154: trip times of cars leaving the map, in road order
      int t = d.n_trips[env];
      for (int x = d.r; x < d.R; ++x) {
        const int idx = env * d.R + x;
        const int rxx = d.rec[idx].x;
        int ps = rec_head(rxx);
        for (int j = 0; j < rec_kpop(rxx); ++j) {
          // the popped car's spawn tick: its ring slot, or row j of the road's outbox (transposed; <= 2 pops here)
          const float cw = !d.w ? 0.0f : (TL ? d.outw[ocol_of(d, env, x) + (size_t)j * 64] : d.w[(size_t)idx * d.C + ps]);
          if (d.trip_times && t < d.trip_cap)
            d.trip_times[(size_t)env * d.trip_cap + t] = side_age(d, tick, cw) / 2.0f;
          ++t;
          ps = wrap1(ps + 1, d.C);
        }
      }
      d.n_trips[env] = t;
    }
  }
}

// GREEDY: the on-device greedy controller is on (the kernels without it keep its loads out of their registers)
template <bool TL, bool HET = false, bool GREEDY = false>
__global__ __launch_bounds__(256) void k_advance(const Dev d, const int tidx) {
  const int tick = *d.tickB;
  const int per_env = d.I + (d.R - d.r);
  const long total = (long)d.E * per_env;
  for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
       gid += (long)gridDim.x * blockDim.x) {
    const int env = (int)(gid / per_env);
    if (gid == 0) *d.tickA = tick + 1;
    advance_item<TL, HET, GREEDY, true, TL>(d, env, (int)(gid - (long)env * per_env), tick, tidx);
  }
}

}  // namespace tfx
