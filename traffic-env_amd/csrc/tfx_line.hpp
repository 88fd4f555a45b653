// tfx_line.hpp - k_line: MANY ticks of the env step per launch, with the cars held on chip.
//
// Cars never turn (roadgraph.py:54-64), so a street line - entry road, the train roads behind it,
// the exit road - exchanges cars and tail positions only with itself; lines meet only at the traffic
// lights, and those are inputs (the action), not state.  With the line-ordered slots of the
// transposed layout (build_slots) a wavefront owns whole lines: lane j holds road j, lane j+1 its
// successor, lane j-1 its predecessor.  k_line therefore
//   1. loads the tile's cars once (the same coalesced row loads as k_move_t) into an LDS ring per
//      road: row (head + k) mod CAPR of column `lane` is the k-th car behind the fake leader;
//   2. runs n_ticks x { lights, spawns, move, handoff } entirely inside the wavefront - the leader
//      chain in registers (k_move_t's walk), the handoff by reading the predecessor lane's popped
//      rows from LDS, tail positions and pop counts through lane shuffles, no barrier anywhere;
//   3. writes cars, ring indices and the per-road outputs back once.
// HBM traffic per tick drops by n_ticks; the kernel is bound by the vector ALU instead.
//
// Exactness: the same idm_step / ring arithmetic as the per-tick kernels, the reference's
// road-order rule for pushes (a push sees the destination's `leading` of before the destination's
// pops iff the source road index is smaller, tfx_advance.hpp) and - instead of the serial fallback
// - cascade rounds: a car handed to an emptied road with a larger index that is still beyond that
// road's end leaves it in the same tick, exactly as the reference's ascending loop does (:117-135).
// Integer outputs, floats and the ring image are bit-identical to tfx_step's per-tick path.
#pragma once
#include "tfx_common.hpp"
#include "tfx_move_t.hpp"

namespace tfx {

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return __builtin_amdgcn_readfirstlane(v);
}

// M = CAPR - 1, CAPR a power of two >= C - 2; dynamic LDS = CAPR * 64 * sizeof(float2)
__global__ __launch_bounds__(64) void k_line(const Dev d, const int tidx0, const int n_ticks, const int M) {
  extern __shared__ float2 ring[];
  const int lane = threadIdx.x;
  const int C = d.C;
  const long tiles = (long)d.E * d.G;
  const int tick0 = *d.tickA;  // advanced by k_tick_add after this kernel, never inside it
  unsigned long long my_updates = 0;

  for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int env = (int)(tile / d.G);
    const int e_slot = d.slot_road[(int)(tile - (long)env * d.G) * 64 + lane];
    const bool valid = e_slot >= 0;
    const int e = valid ? e_slot : 0;
    const int id = env * d.R + e;
    const bool train = valid && e < d.r;
    const int dir = train ? e / d.I : 0;
    const int isec = e - dir * d.I;          // dest intersection of a train road
    const int phase_e = (dir < 2) ? 1 : 0;   // roadgraph.py:36
    const int ej = valid ? d.entry_idx[e] : -1;
    const int pe = valid ? d.pred[e] : -1;
    const bool has_pred = pe >= 0;           // the predecessor is lane - 1
    const bool pred_lower = has_pred && pe < e;

    float2 *col = d.xv + ((size_t)tile * d.trows) * 64 + lane;
    int *ob = d.obs + (size_t)env * d.obs_len;

    int ld = valid ? d.leading[id] : 1, lc = valid ? d.lastcar[id] : 1;
    int n = ring_count(ld, lc, C);
    float tail_x = valid ? d.tailx[id] : 0.0f;
    int ph = train ? ob[2 * d.r + isec] : 0, el = train ? ob[2 * d.r + d.I + isec] : 0;
    int head = 0;

    {  // cars -> LDS, position k in row k
      const int kmax = wave_max_i(n);
      for (int k = 0; k < kmax; ++k)
        if (k < n) {
          const f2v t = __builtin_nontemporal_load(reinterpret_cast<const f2v *>(col + (size_t)k * 64));
          ring[k * 64 + lane] = make_float2(t.x, t.y);
        }
    }

    int wait_acc = 0, det = 0, passed_last = 0, ovf_last = 0, last_ovf_tick = -1;
    bool det_set = false, any_pop = false;
    float xL = INFINITY;

    for (int t = 0; t < n_ticks; ++t) {
      const int tick = tick0 + t, tidx = tidx0 + t;
      // ---- lights (TrafficEnv._step :225-232, update_lights :81-94) ---------------------------
      const int n_next = __shfl_down(n, 1, 64);
      const float tx_next = __shfl_down(tail_x, 1, 64);
      xL = INFINITY;
      if (train) {
        int ph_new, el_new;
        light_next(d, env, isec, tick, tidx, ph, el, ph_new, el_new);
        ph = ph_new;
        el = el_new;
        if (phase_e == ph || el < d.yellow) xL = d.length;
        else if (n_next > 0) xL = tx_next + d.length;
      }
      // ---- spawns (add_new_cars :274-283 -> add_car :97-114) ----------------------------------
      const int ld0 = ld;
      int n_tot = n, ovf = 0;
      if (ej >= 0) {
        const int tick_sp = (d.spawn_mode == TFX_SPAWN_PERIODIC) ? tick % d.spawn_period : 0;
        const int c = spawn_count(d, env, e, ej, tick_sp, tidx);
        for (int q = 0; q < c; ++q) {
          const int pos = wrap1(lc + 1, C);
          const float start = (lc != ld) ? (tail_x - d.car_l) - d.car_s0 : INFINITY;
          if (pos != ld) {
            const float xs = (start < 0.0f) ? start : 0.0f;  // min(car.x = 0, start)
            ring[((head + n_tot) & M) * 64 + lane] = make_float2(xs, d.car_v);
            ++n_tot;
            lc = pos;
            tail_x = xs;
          } else {
            ++ovf;
          }
        }
      }
      // ---- move_cars (:187-212): the leader chain of k_move_t over the LDS ring ----------------
      int kpop = 0, n_wait = 0, n_det = 0, slot = ld;
      bool open = true;
      {
        float xprev = xL, vprev = 0.0f, llv = 0.0f;
        const int lc_seg2 = (ld > lc) ? lc : 0;  // wrapped ring: x, not v, is tested on 1..lastcar (:210)
        const int kmax = wave_max_i(n_tot);
        float2 nxt = (n_tot > 0) ? ring[(head & M) * 64 + lane] : make_float2(0.0f, 0.0f);
        for (int k = 0; k < kmax; ++k) {
          const float2 cur = nxt;
          if (k + 1 < n_tot) nxt = ring[((head + k + 1) & M) * 64 + lane];
          if (k < n_tot) {
            float xn, vn;
            const bool off_domain = __builtin_amdgcn_ballot_w64(!idm_fast_domain(cur.y)) != 0ull;
            if (d.fastdiv && !off_domain) idm_step_fast(d, cur.x, cur.y, xprev, vprev, llv, xn, vn);
            else idm_step(d, cur.x, cur.y, xprev, vprev, llv, xn, vn);
            ring[((head + k) & M) * 64 + lane] = make_float2(xn, vn);
            xprev = cur.x;  // OLD state leads the next car (Jacobi)
            vprev = cur.y;
            llv = d.car_l;
            slot = (slot + 1 >= C) ? 1 : slot + 1;
            const bool pop = open && (xn > d.length);  // the while loop of :123
            open = pop;
            kpop += pop ? 1 : 0;
            const float wq = (slot <= lc_seg2) ? xn : vn;
            n_wait += (wq < d.thresh) ? 1 : 0;
            n_det += (xn > d.near_end) ? 1 : 0;
            tail_x = xn;
          }
        }
      }
      if (train && n_tot > 0) {
        wait_acc += n_wait;
        det = n_det;
        det_set = true;
      }
      my_updates += (unsigned long long)(valid ? n_tot : 0);
      // ---- advance_finished_cars (:117-135): pull-form handoff, cascade rounds -----------------
      int passed = 0;
      int n_cur = n_tot;  // cars counted from `head`, including the ones this road is about to pop
      int k_cur = kpop;
      for (;;) {
        int k_p = __shfl_up(k_cur, 1, 64);
        const int head_p = __shfl_up(head, 1, 64);
        if (!has_pred) k_p = 0;
        const int n_before = n_cur;
        const int ld_post = ring_adv(ld, k_cur, C);
        const int ld_seen = pred_lower ? ld0 : ld_post;
        const int kpmax = wave_max_i(k_p);
        for (int j = 0; j < kpmax; ++j) {
          float2 car = make_float2(0.0f, 0.0f);
          if (j < k_p) car = ring[((head_p + j) & M) * 64 + lane - 1];
          __builtin_amdgcn_wave_barrier();  // every lane has read before any lane re-uses a popped row
          if (j < k_p) {
            const float xc = car.x - d.length;  // state[e,xi,newlead] -= length (:130)
            const int pos = wrap1(lc + 1, C);
            const float start = (lc != ld_seen) ? (tail_x - d.car_l) - d.car_s0 : INFINITY;
            if (pos != ld_seen) {
              const float xs = (start < xc) ? start : xc;
              ring[((head + n_cur) & M) * 64 + lane] = make_float2(xs, car.y);
              ++n_cur;
              lc = pos;
              tail_x = xs;
            } else {
              ++ovf;
            }
          }
        }
        head = (head + k_cur) & M;
        n_cur -= k_cur;
        if (k_cur > 0) ld = ld_post;
        if (train) passed += k_cur;
        // a road with a lower-indexed predecessor is visited after it by the reference's loop: cars
        // pushed onto it once it is empty are candidates of its own while loop in the same tick
        int k_next = 0;
        if (pred_lower && n_before == k_cur && n_cur > 0) {
          while (k_next < n_cur && ring[((head + k_next) & M) * 64 + lane].x > d.length) ++k_next;
        }
        k_cur = k_next;
        if (__builtin_amdgcn_ballot_w64(k_cur > 0) == 0ull) break;
      }
      n = n_cur;
      passed_last = passed;
      any_pop = any_pop || (passed > 0);
      ovf_last = ovf;
      if (ovf > 0) last_ovf_tick = tick;
    }

    // ---- write back -----------------------------------------------------------------------------
    {
      const int kmax = wave_max_i(n);
      for (int k = 0; k < kmax; ++k)
        if (k < n) {
          const float2 c2 = ring[((head + k) & M) * 64 + lane];
          f2v t;
          t.x = c2.x;
          t.y = c2.y;
          __builtin_nontemporal_store(t, reinterpret_cast<f2v *>(col + (size_t)k * 64));
        }
    }
    if (valid) {
      d.leading[id] = ld;
      d.lastcar[id] = lc;
      d.tailx[id] = tail_x;
      d.leadx[id] = xL;
      if (train) {
        ob[e] = passed_last;
        if (det_set) ob[d.r + e] = det;
        if (wait_acc) d.waiting[(size_t)env * d.r + e] += wait_acc;
        if (any_pop) d.passed_dst[(size_t)env * d.I + isec] = 1;
        // (phase / elapsed are NOT stored here: the other three lines through this intersection
        // may belong to tiles that have not started yet and must still read the old values;
        // k_line_lights advances them after the launch)
        if (ovf_last > 0) {  // rewards were zeroed before the launch (:233); -= OVERFLOW_PENALTY per drop
          float pen = 0.0f;
          for (int j = 0; j < ovf_last; ++j) pen -= d.ovf_pen;
          atomicAdd(&d.rewards[(size_t)env * d.I + isec], pen);
        }
      }
      if (last_ovf_tick >= 0) atomicMax(&d.done_tick[env], last_ovf_tick + 1);
    }
    __builtin_amdgcn_wave_barrier();
  }

  for (int off = 32; off > 0; off >>= 1) my_updates += __shfl_down(my_updates, off);
  if (lane == 0 && my_updates) veh_add(d.veh, my_updates);
}

// phase and elapsed of every intersection after the n ticks k_line just ran (the same light_next
// sequence its lanes evaluated in registers); runs after k_line, before k_tick_add
__global__ void k_line_lights(const Dev d, const int tidx0, const int n) {
  const int tick0 = *d.tickA;
  const long total = (long)d.E * d.I;
  for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
       gid += (long)gridDim.x * blockDim.x) {
    const int env = (int)(gid / d.I);
    const int i = (int)(gid - (long)env * d.I);
    int *ob = d.obs + (size_t)env * d.obs_len + 2 * d.r;
    int ph = ob[i], el = ob[d.I + i];
    for (int t = 0; t < n; ++t) {
      int ph_new, el_new;
      light_next(d, env, i, tick0 + t, tidx0 + t, ph, el, ph_new, el_new);
      ph = ph_new;
      el = el_new;
    }
    ob[i] = ph;
    ob[d.I + i] = el;
  }
}

// the tick counters after a fused launch (k_move_t / k_advance keep them per tick)
__global__ void k_tick_add(const Dev d, const int n) {
  const int t = *d.tickA + n;
  *d.tickA = t;
  *d.tickB = t - 1;
}

}  // namespace tfx
