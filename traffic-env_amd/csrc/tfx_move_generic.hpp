// tfx_move_generic.hpp - k_move<WPR>: the general form of the move kernel.  One road per group of
// WPR wavefronts (64*WPR >= C-2 car lanes), any ring capacity up to 258.  Used for roads longer
// than one wavefront (C-2 > 64), for odd capacities, and as the A/B baseline of k_move_dma.
//
// Per road: light phase update, spawn pushes, fake leader (update_lights :81-94), IDM over every
// live car (sim :50-62 / move_cars :187-212), waiting/detected counts, and the number of cars that
// crossed the road end (the pop prefix of advance_finished_cars :123) from wave ballots.  Lane k
// owns the k-th car behind the fake leader; the leader's (x, v) reach the follower through an LDS
// tile: cars are staged at index k+1, the fake leader at index 0, every lane reads index k.
#pragma once
#include "tfx_common.hpp"

namespace tfx {

template <int WPR>
__global__ __launch_bounds__(256) void k_move(const Dev d, const int tidx) {
  constexpr int CPR = 64 * WPR;   // car lanes per road
  constexpr int RPB = 256 / CPR;  // roads per block pass
  __shared__ float sx[RPB][CPR + 1];
  __shared__ float sv[RPB][CPR + 1];
  __shared__ int s_part[RPB][WPR][6];

  const int tid = threadIdx.x;
  const int lr = tid / CPR;   // road within the block pass (wave-uniform)
  const int k = tid % CPR;    // car index behind the fake leader
  const int wq = k >> 6;      // wave within the road
  const int tick = *d.tickA;
  const int C = d.C;
  const int tick_sp = (d.spawn_mode == TFX_SPAWN_PERIODIC) ? tick % d.spawn_period : 0;

  // XCD-aware placement: blocks b and b+8 share an XCD (round-robin dispatch), so give XCD x the
  // x-th contiguous eighth of the roads - neighbouring roads (shared cache lines, next-road tail
  // reads) then meet in one L2.  Placement only affects speed.
  const int G = gridDim.x;
  const int lb = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const long total = (long)d.E * d.R;
  const long groups = (total + RPB - 1) / RPB;
  const long chunk = (groups + G - 1) / G;
  const long g0 = (long)lb * chunk;
  const long g1 = (g0 + chunk < groups) ? g0 + chunk : groups;

  unsigned long long my_updates = 0;

  for (long grp = g0; grp < g1; ++grp) {
    const long idl = grp * RPB + lr;
    const bool in_range = idl < total;
    const int id = __builtin_amdgcn_readfirstlane((int)(in_range ? idl : 0));
    const int env = id / d.R;
    const bool active = in_range && !env_frozen(d, env, tick);  // a frozen env's roads stand still
    const int e = id - env * d.R;
    const bool train = e < d.r;
    const int dst = train ? e % d.I : 0;

    // ring indices, light state -> fake leader, spawns: evaluated by every lane of the road
    // (wave-uniform loads); lane 0 writes the new lastcar below, after the barrier that orders it
    // behind every wave's read of the old one
    const RoadPrep p = prep_road(d, id, env, e, tick, tick_sp, tidx, active, false);
    const int lc = p.lc, n_tot = p.n_tot, ovf_sp = p.ovf_sp;
    const float xs0 = p.xs0;
    const int ld = p.ld, n_old = p.n_old;
    const float xL = p.xL;

    float2 *rx = d.xv + (size_t)id * C;
    const bool is_live = active && k < n_tot;
    const bool is_spawned = is_live && k >= n_old;
    const int slot = is_live ? ring_adv(ld, 1 + k, C) : 0;
    float x = 0.0f, v = 0.0f;
    if (is_live && !is_spawned) {
      const float2 c = rx[slot];
      x = c.x;
      v = c.y;
    } else if (is_spawned) {
      x = spawned_x(d, xs0, k - n_old);
      v = d.car_v;
    }

    // ---- stage (x, v) in LDS: index 0 = fake leader (v = 0, l = 0), index k+1 = car k ---------
    if (WPR > 1) __syncthreads();  // previous pass finished reading the tile
    if (k == 0) {
      sx[lr][0] = xL;
      sv[lr][0] = 0.0f;
    }
    sx[lr][k + 1] = x;
    sv[lr][k + 1] = v;
    if (WPR > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    if (active && k == 0 && n_tot != n_old) d.lastcar[id] = lc;
    const float xl = sx[lr][k];
    const float vl = sv[lr][k];
    const float ll = (k == 0) ? 0.0f : d.car_l;

    float xn, vn;
    idm_step(d, x, v, xl, vl, ll, xn, vn);

    if (is_live) {
      rx[slot] = make_float2(xn, vn);
      if (is_spawned && d.w) d.w[(size_t)id * C + slot] = (float)tick;
    }
    if (active && k == 0) {
      rx[ld].x = xL;  // the reference keeps the leader's x in its slot
      d.leadx[id] = xL;  // and k_advance re-installs it in the slot a pop frees (advance_road)
    }

    // ---- counts (move_cars :199-201, :208-212) and the pop prefix (:123) ---------------------
    const bool seg2 = (ld > lc) && (slot <= lc);  // wrapped ring, second segment: x tested, not v
    const bool c_wait = is_live && ((seg2 ? xn : vn) < d.thresh);
    const bool c_det = is_live && (xn > d.near_end);
    const bool c_pop = is_live && (xn > d.length);
    const bool c_far = c_pop && ((xn - d.length) > d.length);  // would be popped again downstream
    const unsigned long long m_pop = __builtin_amdgcn_ballot_w64(c_pop);
    int n_wait = __popcll(__builtin_amdgcn_ballot_w64(c_wait));
    int n_det = __popcll(__builtin_amdgcn_ballot_w64(c_det));
    // leading ones of m_pop = cars popped from the head (the while loop stops at the first car
    // that is still on the road)
    int kpop = (~m_pop == 0ull) ? 64 : __builtin_ctzll(~m_pop);
    int any_far = (__builtin_amdgcn_ballot_w64(c_far) != 0ull) ? 1 : 0;
    if (WPR > 1) {
      if ((tid & 63) == 0) {
        s_part[lr][wq][0] = n_wait;
        s_part[lr][wq][1] = n_det;
        s_part[lr][wq][2] = kpop;
        s_part[lr][wq][4] = any_far;
      }
      __syncthreads();
      n_wait = 0; n_det = 0; kpop = 0; any_far = 0;
      bool open = true;
#pragma unroll
      for (int w = 0; w < WPR; ++w) {
        n_wait += s_part[lr][w][0];
        n_det += s_part[lr][w][1];
        if (open) {
          kpop += s_part[lr][w][2];
          open = s_part[lr][w][2] == 64;  // whole wave popped: the prefix continues
        }
        any_far |= s_part[lr][w][4];
      }
    }
    const bool slow = needs_serial(kpop, any_far != 0, n_tot, C);

    if (active) {
      int *ob = d.obs + (size_t)env * d.obs_len;
      if (k == 0) {
        if (train) {
          if (n_tot > 0) {
            d.waiting[(size_t)env * d.r + e] += n_wait;
            ob[d.r + e] = n_det;
          }
          ob[e] = (d.agent_mode && tidx > 0) ? ob[e] + kpop : kpop;
          if (kpop > 0) d.passed_dst[(size_t)env * d.I + dst] = 1;
        }
        int *rp = reinterpret_cast<int *>(d.rec + id);
        rp[0] = rec_pack(kpop, ld, C);
        rp[1] = ovf_sp;
        rp[3] = n_tot;
        if (slow) d.env_flag[env] = tick + 1;
        my_updates += (unsigned long long)n_tot;
      }
      if (is_live && k == n_tot - 1) reinterpret_cast<float *>(d.rec + id)[2] = xn;
    }
  }

  if (my_updates) veh_add(d.veh, my_updates);
  if (blockIdx.x == 0 && tid == 0) *d.tickB = tick;
}

}  // namespace tfx
