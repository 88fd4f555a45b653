// tfx_hip.hip - MI355X (gfx950 / CDNA4) implementation of the IDM traffic-env tick behind the
// C ABI of include/tfx.h.  Written for wave64; no other target is supported.
//
// What one tick does (reference: gym_traffic/envs/traffic_env.py:224-248, TrafficEnv._step):
//   k_move     one wavefront (or WPR wavefronts) per road: light phase update, spawn pushes, fake
//              leader (update_lights :81-94), IDM over every live car (sim :50-62 / move_cars
//              :187-212), waiting/detected counts, and the count of cars that crossed the road end
//              (the pop prefix of advance_finished_cars :123) found with a wave ballot.
//   k_advance  one lane per intersection (its 4 incoming roads) or exit road: ring pop + handoff
//              (advance_finished_cars :117-135 / advance_hack :139-157) in PULL form - each road
//              takes the cars its unique predecessor popped - with the reference's sequential
//              road-order rule reproduced exactly (see advance_road).
// The reference walks roads sequentially; the parallel form is exact whenever every road pops at
// most TFX_KP cars and no handed-off car could be popped again in the same tick.  k_move detects
// the contrary per env and k_advance then runs that env through advance_env_serial, a literal
// single-thread restatement - so results equal the sequential algorithm in every case.
//
// Memory: cars of a road are contiguous (x plane then v plane), lane k of the road's wave(s) owns
// the k-th car behind the fake leader, so loads/stores are coalesced up to the ring wrap.  The
// leader's (x, v) reach the follower through an LDS tile: cars are staged at index k+1, the fake
// leader at index 0, and every lane reads index k - its leader - as a +1-offset LDS access.
// HBM-bound (16 B per vehicle-update); no MFMA: there is no contraction in this path.
//
// Float contract (bit-for-bit shared with oracle/idm_oracle.c): binary32, the reference's
// expression order, one rounding per op (-ffp-contract=off), correctly rounded div/sqrt,
// q^4 via two binary64 multiplies, np.maximum(0,t) as (0 >= t ? 0 : t).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "tfx.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(expr)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) return fail(TFX_EDEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// Everything a kernel needs, passed by value.
struct Dev {
  int I, r, R, C, E, P, n_entry, obs_len;
  int yellow, learn_switch, validate, env_off;
  float length, rate, car_v, car_l, car_a, car_v0, car_b, car_T, car_s0;
  float two_sab, eps, thresh, near_end, ovf_pen;
  // caller-owned state
  float *state;
  int *leading, *lastcar, *obs;
  float *rewards;
  int *waiting;
  uint8_t *passed_dst;
  int *done_tick;
  float *trip_times;
  int *n_trips;
  int trip_cap;
  // handle-owned tables + scratch
  const int *nexts, *pred, *entry_idx;
  int4 *rec;      // per road: {pops k, spawn overflows, bits of post-move tail x, live cars}
  float *popcar;  // per road: TFX_KP x {x, v, w} of the cars popped this tick
  float *tailx;   // per road: x of the last car after the advance (what update_lights reads)
  int *env_flag;  // == tick+1 when the env must take the serial advance this tick
  unsigned long long *veh;
  int *tickA, *tickB;
  // per-tick inputs
  const int *action;
  int action_mode, action_period;
  long action_stride;
  const int *spawn;
  int spawn_mode, spawn_period;
  long spawn_stride;
};

constexpr int KP = TFX_KP;

__device__ __forceinline__ float np_max0(float t) { return (0.0f >= t) ? 0.0f : t; }
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

// TrafficEnv._step lines :225-232 for one intersection: new phase and elapsed from the old ones.
__device__ __forceinline__ void light_update(const Dev &d, int env, int i, int tick, int tidx,
                                             int &ph_new, int &el_new) {
  const int *ob = d.obs + (size_t)env * d.obs_len + 2 * d.r;
  const int ph = ob[i], el = ob[d.I + i];
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

__device__ __forceinline__ int spawn_count(const Dev &d, int env, int e, int ej, int tick, int tidx) {
  if (d.spawn_mode == TFX_SPAWN_COUNTS)
    return d.spawn[(size_t)tidx * d.spawn_stride + (size_t)env * d.n_entry + ej];
  if (d.spawn_mode == TFX_SPAWN_PERIODIC) return (tick % d.spawn_period) == (e % d.spawn_period) ? 1 : 0;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// k_move: lights + spawns + IDM + counts + pop prefix.  WPR = wavefronts per road (64*WPR >= C-2).
// ------------------------------------------------------------------------------------------------
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
    const bool active = idl < total;
    const int id = __builtin_amdgcn_readfirstlane((int)(active ? idl : 0));
    const int env = id / d.R;
    const int e = id - env * d.R;
    const bool train = e < d.r;
    const int dst = train ? e % d.I : -1;

    int ld = 1, lc = 1;
    if (active) {
      ld = d.leading[id];
      lc = d.lastcar[id];
    }
    const int n = ring_count(ld, lc, C);

    float *xs = d.state + ((size_t)id * d.P) * C;
    float *vs = xs + C;
    float *ws = xs + 2 * C;

    // ---- existing cars: lane k owns the k-th car behind the leader --------------------------
    const bool is_old = active && k < n;
    int slot = is_old ? ring_adv(ld, 1 + k, C) : 0;
    float x = 0.0f, v = 0.0f;
    if (is_old) {
      x = xs[slot];
      v = vs[slot];
    }

    // ---- light state of the destination intersection (TrafficEnv._step :225-232) -----------
    int ph_new = 0, el_new = 0;
    if (active && train) light_update(d, env, dst, tick, tidx, ph_new, el_new);

    // ---- spawns onto entry roads (add_new_cars :274-283 -> add_car :97-114) -----------------
    int n_tot = n, ovf_sp = 0;
    bool is_spawned = false;
    const int ej = active ? d.entry_idx[e] : -1;
    if (ej >= 0) {
      const int c = spawn_count(d, env, e, ej, tick, tidx);
      if (c > 0) {
        float tail_x = d.tailx[id];
        for (int j = 0; j < c; ++j) {
          const int pos = wrap1(lc + 1, C);
          const float start = (lc != ld) ? (tail_x - d.car_l) - d.car_s0 : INFINITY;
          if (pos != ld) {
            const float xn = (start < 0.0f) ? start : 0.0f;  // min(car.x = 0, start)
            if (k == n_tot) {
              x = xn;
              v = d.car_v;
              slot = pos;
              is_spawned = true;
            }
            ++n_tot;
            lc = pos;
            tail_x = xn;
          } else {
            ++ovf_sp;
          }
        }
        if (k == 0) d.lastcar[id] = lc;
      }
    }

    // ---- fake leader (update_lights :81-94) --------------------------------------------------
    float xL = INFINITY;
    if (active && train) {
      const int phase_e = (e / d.I < 2) ? 1 : 0;  // roadgraph.py:36
      if (phase_e == ph_new || el_new < d.yellow) {
        xL = d.length;
      } else {
        const int idn = env * d.R + d.nexts[e];
        if (d.lastcar[idn] != d.leading[idn]) xL = d.tailx[idn] + d.length;
      }
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

    const bool is_live = active && k < n_tot;
    const float xl = sx[lr][k];
    const float vl = sv[lr][k];
    const float ll = (k == 0) ? 0.0f : d.car_l;

    // ---- IDM (sim :50-62), evaluated from OLD values only ------------------------------------
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
    const float xn = x + (dx > 0.0f ? dx : 0.0f * dx);
    const float vn = np_max0(v + dvr);

    if (is_live) {
      xs[slot] = xn;
      vs[slot] = vn;
      if (is_spawned && d.P == 3) ws[slot] = (float)tick;
    }
    if (active && k == 0) xs[ld] = xL;  // the reference keeps the leader's x in its slot

    // ---- counts (move_cars :199-201, :208-212) and the pop prefix (:123) ---------------------
    const bool seg2 = (ld > lc) && (slot <= lc);  // wrapped ring, second segment: x tested, not v
    const bool c_wait = is_live && ((seg2 ? xn : vn) < d.thresh);
    const bool c_det = is_live && (xn > d.near_end);
    const bool c_pop = is_live && (xn > d.length);
    const bool c_far = c_pop && ((xn - d.length) > d.length);  // would be popped again downstream
    const unsigned long long m_pop = __ballot(c_pop);
    const unsigned long long m_live = __ballot(is_live);
    int n_wait = __popcll(__ballot(c_wait));
    int n_det = __popcll(__ballot(c_det));
    // leading ones of m_pop = cars popped from the head (the while loop stops at the first car
    // that is still on the road)
    int kpop = (~m_pop == 0ull) ? 64 : __builtin_ctzll(~m_pop);
    int any_far = (__ballot(c_far) != 0ull) ? 1 : 0;
    if (WPR > 1) {
      const int lane = tid & 63;
      if (lane == 0) {
        s_part[lr][wq][0] = n_wait;
        s_part[lr][wq][1] = n_det;
        s_part[lr][wq][2] = kpop;
        s_part[lr][wq][3] = __popcll(m_live);
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

    // a far car only matters if it is inside the popped prefix; c_pop beyond the prefix cannot
    // happen physically, keep the exact test cheap: flag conservatively
    const bool slow = (kpop > KP) || any_far;

    if (active) {
      int *ob = d.obs + (size_t)env * d.obs_len;
      if (k == 0) {
        if (train) {
          if (n_tot > 0) {
            d.waiting[(size_t)env * d.r + e] += n_wait;
            ob[d.r + e] = n_det;
          }
          ob[e] = kpop;
          if (kpop > 0) d.passed_dst[(size_t)env * d.I + dst] = 1;
        }
        int4 rc;
        rc.x = kpop;
        rc.y = ovf_sp;
        rc.z = 0;
        rc.w = n_tot;
        int *rp = reinterpret_cast<int *>(d.rec + id);
        rp[0] = rc.x;
        rp[1] = rc.y;
        rp[3] = rc.w;
        if (slow) d.env_flag[env] = tick + 1;
        my_updates += (unsigned long long)n_tot;
      }
      if (is_live && k == n_tot - 1) reinterpret_cast<float *>(d.rec + id)[2] = xn;
      if (is_live && k < kpop && k < KP) {
        float w = 0.0f;
        if (d.P == 3) w = is_spawned ? (float)tick : ws[slot];
        float *pc = d.popcar + ((size_t)id * KP + k) * 3;
        pc[0] = xn;
        pc[1] = vn;
        pc[2] = w;
      }
    }
  }

  if (my_updates) atomicAdd(d.veh, my_updates);
  if (blockIdx.x == 0 && tid == 0) *d.tickB = tick;
}

// ------------------------------------------------------------------------------------------------
// k_move_tile: the C-2 <= 64 form of k_move (one wavefront per road), restructured so that a wave
// owns a TILE of 64 consecutive roads and works in three wave-local phases - no block barrier:
//   M  lane j prepares road j of the tile: ring indices, light state, fake-leader x, spawns.  All
//      per-road scalar work (index arithmetic, table/obs gathers, the next road's tail) is done
//      64 roads at a time with coalesced loads instead of once per road with wave-uniform loads.
//   C  for each road of the tile, all 64 lanes take one car each: coalesced x/v loads, LDS leader
//      staging, IDM, coalesced stores, ballot counts.  Loads of the next group of U roads are
//      issued before the current group is computed, so 2*U roads are in flight per wave.
//   W  lane j writes road j's results (waiting/detected/passed, the handoff record) coalesced.
// Descriptors and results travel through LDS entries that only this wave touches.
// ------------------------------------------------------------------------------------------------
struct RoadDesc {
  int ld, lc, n_old, n_tot;
  float xL, xs0;
  int ovf_sp, pad;
};

struct CarRegs {
  float x, v;
  int slot;
};

__device__ __forceinline__ void tile_load(const Dev &d, const RoadDesc &sd, int id, int lane, CarRegs &c) {
  const int C = d.C;
  const float *xs = d.state + ((size_t)id * d.P) * C;
  c.slot = ring_adv(sd.ld, 1 + lane, C);
  c.x = 0.0f;
  c.v = 0.0f;
  if (lane < sd.n_old) {
    c.x = xs[c.slot];
    c.v = xs[C + c.slot];
  } else if (lane < sd.n_tot) {
    // car spawned this tick: the j-th accepted spawn queues behind the (j-1)-th (add_car :100-107)
    float xv = sd.xs0;
    for (int j = sd.n_old; j < lane; ++j) xv = (xv - d.car_l) - d.car_s0;
    c.x = xv;
    c.v = d.car_v;
  }
}

__device__ __forceinline__ void tile_compute(const Dev &d, const RoadDesc &sd, int id, int lane, int tick,
                                             float *sx, float *sv, int4 *res, const CarRegs &c) {
  const int C = d.C;
  float *xs = d.state + ((size_t)id * d.P) * C;
  float *vs = xs + C;
  float *ws = xs + 2 * C;
  const float x = c.x, v = c.v;
  // stage: index 0 = fake leader (v = 0, l = 0), index k+1 = car k; lane k then reads index k
  if (lane == 0) {
    sx[0] = sd.xL;
    sv[0] = 0.0f;
  }
  sx[lane + 1] = x;
  sv[lane + 1] = v;
  __builtin_amdgcn_wave_barrier();
  const float xl = sx[lane];
  const float vl = sv[lane];
  __builtin_amdgcn_wave_barrier();
  const float ll = (lane == 0) ? 0.0f : d.car_l;

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
  const float xn = x + (dx > 0.0f ? dx : 0.0f * dx);
  const float vn = np_max0(v + dvr);

  const bool is_live = lane < sd.n_tot;
  const bool is_spawned = is_live && lane >= sd.n_old;
  if (is_live) {
    xs[c.slot] = xn;
    vs[c.slot] = vn;
    if (is_spawned && d.P == 3) ws[c.slot] = (float)tick;
  }
  const bool seg2 = (sd.ld > sd.lc) && (c.slot <= sd.lc);
  const bool c_wait = is_live && ((seg2 ? xn : vn) < d.thresh);
  const bool c_det = is_live && (xn > d.near_end);
  const bool c_pop = is_live && (xn > d.length);
  const bool c_far = c_pop && ((xn - d.length) > d.length);
  const unsigned long long m_pop = __ballot(c_pop);
  const int n_wait = __popcll(__ballot(c_wait));
  const int n_det = __popcll(__ballot(c_det));
  const int kpop = (~m_pop == 0ull) ? 64 : __builtin_ctzll(~m_pop);
  const bool slow = (kpop > KP) || (__ballot(c_far) != 0ull);
  if (lane == 0) {
    res->x = kpop | (slow ? (1 << 30) : 0);
    res->y = n_wait;
    res->z = n_det;
  }
  if (is_live && lane == sd.n_tot - 1) res->w = __float_as_int(xn);
  if (is_live && lane < kpop && lane < KP) {
    float w = 0.0f;
    if (d.P == 3) w = is_spawned ? (float)tick : ws[c.slot];
    float *pc = d.popcar + ((size_t)id * KP + lane) * 3;
    pc[0] = xn;
    pc[1] = vn;
    pc[2] = w;
  }
}

template <int U>
__global__ __launch_bounds__(256) void k_move_tile(const Dev d, const int tidx) {
  constexpr int TR = 64;  // roads per wave tile
  __shared__ RoadDesc s_desc[4][TR];
  __shared__ int4 s_res[4][TR];
  __shared__ float s_x[4][TR + 2];
  __shared__ float s_v[4][TR + 2];

  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int tick = *d.tickA;
  const int C = d.C;
  RoadDesc *desc = s_desc[wv];
  int4 *res = s_res[wv];
  float *sx = s_x[wv], *sv = s_v[wv];

  // waves take contiguous runs of tiles; blocks b and b+8 share an XCD, so XCD x gets the x-th
  // eighth of all tiles (neighbouring roads -> one L2).  Speed only.
  const int G = gridDim.x;
  const int lb = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const long total = (long)d.E * d.R;
  const long tiles = (total + TR - 1) / TR;
  const long nw = (long)G * 4;
  const long chunk = (tiles + nw - 1) / nw;
  const long t0 = ((long)lb * 4 + wv) * chunk;
  const long t1 = (t0 + chunk < tiles) ? t0 + chunk : tiles;
  const int tick_sp = (d.spawn_mode == TFX_SPAWN_PERIODIC) ? tick % d.spawn_period : 0;

  unsigned long long my_updates = 0;

  for (long tile = t0; tile < t1; ++tile) {
    const long base = tile * TR;
    // ================= phase M: lane j <-> road base + j ========================================
    const bool valid = base + lane < total;
    const int id = valid ? (int)(base + lane) : (int)(total - 1);
    const int env = id / d.R;
    const int e = id - env * d.R;
    const bool train = e < d.r;
    const int dir = train ? e / d.I : 0;
    const int dst = e - dir * d.I;
    RoadDesc rd;
    rd.ld = d.leading[id];
    rd.lc = d.lastcar[id];
    rd.n_old = ring_count(rd.ld, rd.lc, C);
    rd.n_tot = rd.n_old;
    rd.xL = INFINITY;
    rd.xs0 = 0.0f;
    rd.ovf_sp = 0;
    rd.pad = 0;
    if (train) {
      int ph_new, el_new;
      light_update(d, env, dst, tick, tidx, ph_new, el_new);
      const int phase_e = (dir < 2) ? 1 : 0;
      if (phase_e == ph_new || el_new < d.yellow) {
        rd.xL = d.length;
      } else {
        const int idn = env * d.R + d.nexts[e];
        if (d.lastcar[idn] != d.leading[idn]) rd.xL = d.tailx[idn] + d.length;
      }
    }
    const int ej = d.entry_idx[e];
    if (ej >= 0 && valid) {
      int c;
      if (d.spawn_mode == TFX_SPAWN_COUNTS)
        c = d.spawn[(size_t)tidx * d.spawn_stride + (size_t)env * d.n_entry + ej];
      else if (d.spawn_mode == TFX_SPAWN_PERIODIC)
        c = (tick_sp == e % d.spawn_period) ? 1 : 0;
      else
        c = 0;
      if (c > 0) {
        float tail_x = d.tailx[id];
        for (int j = 0; j < c; ++j) {
          const int pos = wrap1(rd.lc + 1, C);
          const float start = (rd.lc != rd.ld) ? (tail_x - d.car_l) - d.car_s0 : INFINITY;
          if (pos != rd.ld) {
            const float xv = (start < 0.0f) ? start : 0.0f;
            if (rd.n_tot == rd.n_old) rd.xs0 = xv;
            ++rd.n_tot;
            rd.lc = pos;
            tail_x = xv;
          } else {
            ++rd.ovf_sp;
          }
        }
        if (rd.n_tot != rd.n_old) d.lastcar[id] = rd.lc;
      }
    }
    if (valid) d.state[((size_t)id * d.P) * C + rd.ld] = rd.xL;  // the leader's x stays in its slot
    desc[lane] = rd;
    __builtin_amdgcn_wave_barrier();

    // ================= phase C: all lanes on one road at a time ==================================
    const long left = total - base;
    const int nroads = left < TR ? (int)left : TR;
    const int idb = (int)base;
    CarRegs ra[U], rb[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (u < nroads) tile_load(d, desc[u], idb + u, lane, ra[u]);
    for (int g = 0; g < nroads; g += 2 * U) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (g + U + u < nroads) tile_load(d, desc[g + U + u], idb + g + U + u, lane, rb[u]);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (g + u < nroads) tile_compute(d, desc[g + u], idb + g + u, lane, tick, sx, sv, &res[g + u], ra[u]);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (g + 2 * U + u < nroads) tile_load(d, desc[g + 2 * U + u], idb + g + 2 * U + u, lane, ra[u]);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (g + U + u < nroads)
          tile_compute(d, desc[g + U + u], idb + g + U + u, lane, tick, sx, sv, &res[g + U + u], rb[u]);
    }
    __builtin_amdgcn_wave_barrier();

    // ================= phase W: lane j writes road j's results ===================================
    if (valid) {
      const int4 rs = res[lane];
      const int kpop = rs.x & 0xffff;
      const bool slow = (rs.x >> 30) & 1;
      if (train) {
        int *ob = d.obs + (size_t)env * d.obs_len;
        if (rd.n_tot > 0) {
          d.waiting[(size_t)env * d.r + e] += rs.y;
          ob[d.r + e] = rs.z;
        }
        ob[e] = kpop;
        if (kpop > 0) d.passed_dst[(size_t)env * d.I + dst] = 1;
      }
      d.rec[id] = make_int4(kpop, rd.ovf_sp, rs.w, rd.n_tot);
      if (slow) d.env_flag[env] = tick + 1;
      my_updates += (unsigned long long)rd.n_tot;
    }
    __builtin_amdgcn_wave_barrier();
  }

  // one atomic per wave
  for (int off = 32; off > 0; off >>= 1) my_updates += __shfl_down(my_updates, off);
  if (lane == 0 && my_updates) atomicAdd(d.veh, my_updates);
  if (blockIdx.x == 0 && threadIdx.x == 0) *d.tickB = tick;
}

// ------------------------------------------------------------------------------------------------
// k_move_w64: k_move_tile with the per-road scalars kept out of memory altogether.
//   * the lane that prepared road j in phase M keeps its descriptor in registers; phase C fetches
//     it with v_readlane (-> SGPRs), so ring indices, car counts and the leader's x are scalar
//     values, every branch on them is wave-uniform, and no descriptor LDS traffic remains;
//   * results go the other way: lane j picks up road j's ballot counts with a select;
//   * car addresses are a uniform tile base (SGPR pair) + a 32-bit lane offset;
//   * the leader's (x, v) reach the follower either through the LDS tile (+1 read, LEADER_LDS) or
//     through a DPP wave shift-right-by-one that injects the fake leader into lane 0 (no LDS);
//     bench.py A/Bs the two (TFX_MOVE_VARIANT) - the result is identical bit for bit.
// ------------------------------------------------------------------------------------------------
struct CarR {
  float x, v;
  int off;  // float index of the car's x relative to the tile base
};

__device__ __forceinline__ int pack_desc(int ld, int lc, int n_old, int n_tot) {
  return ld | (lc << 9) | (n_old << 18) | (n_tot << 25);
}

template <bool LEADER_LDS>
__device__ __forceinline__ void w64_load(const Dev &d, const float *tx, int j, int pkj, int xs0_bits,
                                         int lane, CarR &c) {
  const int C = d.C;
  const int ld = pkj & 511, n_old = (pkj >> 18) & 127, n_tot = (int)((unsigned)pkj >> 25);
  int slot = ld + 1 + lane;
  slot = (slot >= C) ? slot - (C - 1) : slot;
  c.off = j * (d.P * C) + slot;
  c.x = 0.0f;
  c.v = 0.0f;
  if (lane < n_old) {
    c.x = tx[c.off];
    c.v = tx[c.off + C];
  }
  if (n_tot != n_old) {  // wave-uniform: cars spawned on this road this tick
    if (lane >= n_old && lane < n_tot) {
      float xv = __int_as_float(xs0_bits);
      for (int q = n_old; q < lane; ++q) xv = (xv - d.car_l) - d.car_s0;
      c.x = xv;
      c.v = d.car_v;
    }
  }
}

template <bool LEADER_LDS>
__device__ __forceinline__ void w64_compute(const Dev &d, float *tx, int idb, int j, int pkj, float xL,
                                            int lane, int tick, float *sx, float *sv, const CarR &c,
                                            int &r_k, int &r_w, int &r_d, int &r_t) {
  const int C = d.C;
  const int ld = pkj & 511, lc = (pkj >> 9) & 511, n_old = (pkj >> 18) & 127;
  const int n_tot = (int)((unsigned)pkj >> 25);
  const float x = c.x, v = c.v;
  float xl, vl;
  if (LEADER_LDS) {
    if (lane == 0) {
      sx[0] = xL;
      sv[0] = 0.0f;
    }
    sx[lane + 1] = x;
    sv[lane + 1] = v;
    __builtin_amdgcn_wave_barrier();
    xl = sx[lane];
    vl = sv[lane];
    __builtin_amdgcn_wave_barrier();
  } else {
    // wave_shr:1 - lane k receives lane k-1, lane 0 keeps `old` = the fake leader (x = xL, v = 0)
    xl = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(xL), __float_as_int(x), 0x138, 0xf, 0xf, false));
    vl = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
  }
  const float ll = (lane == 0) ? 0.0f : d.car_l;

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
  const float xn = x + (dx > 0.0f ? dx : 0.0f * dx);
  const float vn = np_max0(v + dvr);

  const bool is_live = lane < n_tot;
  if (is_live) {
    tx[c.off] = xn;
    tx[c.off + C] = vn;
  }
  if (n_tot != n_old && d.P == 3) {
    if (lane >= n_old && lane < n_tot) tx[c.off + 2 * C] = (float)tick;
  }
  bool c_wait;
  if (ld > lc) {  // wrapped ring: the reference tests x, not v, on the second segment (:210)
    const int slot = c.off - j * (d.P * C);
    c_wait = ((slot <= lc) ? xn : vn) < d.thresh;
  } else {
    c_wait = vn < d.thresh;
  }
  const unsigned long long m_pop = __ballot(is_live && (xn > d.length));
  const int n_wait = __popcll(__ballot(is_live && c_wait));
  const int n_det = __popcll(__ballot(is_live && (xn > d.near_end)));
  int kpop = (~m_pop == 0ull) ? 64 : __builtin_ctzll(~m_pop);
  int tail_bits = 0;
  if (n_tot > 0) tail_bits = __builtin_amdgcn_readlane(__float_as_int(xn), n_tot - 1);
  if (kpop > 0) {  // wave-uniform, ~1 road-tick in 10
    const bool far = is_live && (xn > d.length) && ((xn - d.length) > d.length);
    const bool slow = (kpop > KP) || (__ballot(far) != 0ull);
    if (lane < kpop && lane < KP) {
      float w = 0.0f;
      if (d.P == 3) w = (lane >= n_old) ? (float)tick : tx[c.off + 2 * C];
      float *pc = d.popcar + ((size_t)(idb + j) * KP + lane) * 3;
      pc[0] = xn;
      pc[1] = vn;
      pc[2] = w;
    }
    kpop |= slow ? (1 << 30) : 0;
  }
  const bool mine = lane == j;  // lane j collects road j's results
  r_k = mine ? kpop : r_k;
  r_w = mine ? n_wait : r_w;
  r_d = mine ? n_det : r_d;
  r_t = mine ? tail_bits : r_t;
}

template <int U, bool LEADER_LDS>
__global__ __launch_bounds__(256) void k_move_w64(const Dev d, const int tidx) {
  constexpr int TR = 64;
  __shared__ float s_x[4][TR + 2];
  __shared__ float s_v[4][TR + 2];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int tick = *d.tickA;
  const int C = d.C;
  float *sx = s_x[wv], *sv = s_v[wv];

  const int G = gridDim.x;
  const int lb = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const long total = (long)d.E * d.R;
  const long tiles = (total + TR - 1) / TR;
  const long nw = (long)G * 4;
  const long chunk = (tiles + nw - 1) / nw;
  const long t0 = ((long)lb * 4 + wv) * chunk;
  const long t1 = (t0 + chunk < tiles) ? t0 + chunk : tiles;
  const int tick_sp = (d.spawn_mode == TFX_SPAWN_PERIODIC) ? tick % d.spawn_period : 0;

  unsigned long long my_updates = 0;

  for (long tile = t0; tile < t1; ++tile) {
    const long base = tile * TR;
    // ================= phase M: lane j <-> road base + j ========================================
    const bool valid = base + lane < total;
    const int id = valid ? (int)(base + lane) : (int)(total - 1);
    const int env = id / d.R;
    const int e = id - env * d.R;
    const bool train = e < d.r;
    const int dir = train ? e / d.I : 0;
    const int dst = e - dir * d.I;
    const int ld = d.leading[id];
    int lc = d.lastcar[id];
    const int n_old = ring_count(ld, lc, C);
    int n_tot = n_old, ovf_sp = 0;
    float xL = INFINITY, xs0 = 0.0f;
    if (train) {
      int ph_new, el_new;
      light_update(d, env, dst, tick, tidx, ph_new, el_new);
      const int phase_e = (dir < 2) ? 1 : 0;
      if (phase_e == ph_new || el_new < d.yellow) {
        xL = d.length;
      } else {
        const int idn = env * d.R + d.nexts[e];
        if (d.lastcar[idn] != d.leading[idn]) xL = d.tailx[idn] + d.length;
      }
    }
    const int ej = d.entry_idx[e];
    if (ej >= 0 && valid) {
      int c;
      if (d.spawn_mode == TFX_SPAWN_COUNTS)
        c = d.spawn[(size_t)tidx * d.spawn_stride + (size_t)env * d.n_entry + ej];
      else if (d.spawn_mode == TFX_SPAWN_PERIODIC)
        c = (tick_sp == e % d.spawn_period) ? 1 : 0;
      else
        c = 0;
      if (c > 0) {
        float tail_x = d.tailx[id];
        for (int q = 0; q < c; ++q) {
          const int pos = wrap1(lc + 1, C);
          const float start = (lc != ld) ? (tail_x - d.car_l) - d.car_s0 : INFINITY;
          if (pos != ld) {
            const float xv = (start < 0.0f) ? start : 0.0f;
            if (n_tot == n_old) xs0 = xv;
            ++n_tot;
            lc = pos;
            tail_x = xv;
          } else {
            ++ovf_sp;
          }
        }
        if (n_tot != n_old) d.lastcar[id] = lc;
      }
    }
    if (valid) d.state[((size_t)id * d.P) * C + ld] = xL;  // the leader's x stays in its slot
    const int pk = pack_desc(ld, lc, n_old, n_tot);
    const int xL_bits = __float_as_int(xL), xs0_bits = __float_as_int(xs0);

    // ================= phase C: all lanes on one road at a time ==================================
    const long left = total - base;
    const int nroads = left < TR ? (int)left : TR;
    const int idb = (int)base;
    float *tx = d.state + (size_t)base * d.P * C;  // wave-uniform tile base
    int r_k = 0, r_w = 0, r_d = 0, r_t = 0;
    CarR ra[U], rb[U];
#define TFX_LOAD(J, REG)                                                                          \
  w64_load<LEADER_LDS>(d, tx, (J), __builtin_amdgcn_readlane(pk, (J)),                             \
                       __builtin_amdgcn_readlane(xs0_bits, (J)), lane, (REG))
#define TFX_COMPUTE(J, REG)                                                                       \
  w64_compute<LEADER_LDS>(d, tx, idb, (J), __builtin_amdgcn_readlane(pk, (J)),                     \
                          __int_as_float(__builtin_amdgcn_readlane(xL_bits, (J))), lane, tick, sx, \
                          sv, (REG), r_k, r_w, r_d, r_t)
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (u < nroads) TFX_LOAD(u, ra[u]);
    for (int g = 0; g < nroads; g += 2 * U) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (g + U + u < nroads) TFX_LOAD(g + U + u, rb[u]);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (g + u < nroads) TFX_COMPUTE(g + u, ra[u]);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (g + 2 * U + u < nroads) TFX_LOAD(g + 2 * U + u, ra[u]);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (g + U + u < nroads) TFX_COMPUTE(g + U + u, rb[u]);
    }
#undef TFX_LOAD
#undef TFX_COMPUTE

    // ================= phase W: lane j writes road j's results ===================================
    if (valid) {
      const int kpop = r_k & 0xffff;
      if (train) {
        int *ob = d.obs + (size_t)env * d.obs_len;
        if (n_tot > 0) {
          d.waiting[(size_t)env * d.r + e] += r_w;
          ob[d.r + e] = r_d;
        }
        ob[e] = kpop;
        if (kpop > 0) d.passed_dst[(size_t)env * d.I + dst] = 1;
      }
      d.rec[id] = make_int4(kpop, ovf_sp, r_t, n_tot);
      if ((r_k >> 30) & 1) d.env_flag[env] = tick + 1;
      my_updates += (unsigned long long)n_tot;
    }
  }

  for (int off = 32; off > 0; off >>= 1) my_updates += __shfl_down(my_updates, off);
  if (lane == 0 && my_updates) atomicAdd(d.veh, my_updates);
  if (blockIdx.x == 0 && threadIdx.x == 0) *d.tickB = tick;
}

// ------------------------------------------------------------------------------------------------
// k_move_dma: the HBM-facing form of the tick.  Same three wave-local phases as k_move_w64, but
// phase C brings the road records in with LDS-DMA (global_load_lds_dwordx4): a sub-tile of S
// consecutive roads is one contiguous, 16-byte-aligned span of S*P*C floats, copied to LDS by
// whole-wave instructions of 64 lanes x 16 B = 1 KiB - the widest, fully coalesced access the
// memory system has, with no VGPR holding the bytes in flight, so each waiting wave keeps
// S*P*C*4 bytes (4.1 KiB at C = 66) outstanding instead of two dwords per lane.  Cars are then
// read from the LDS image by ring slot; the follower finds its leader either at the previous ring
// slot of the image (the +1 LDS access; the fake leader's (x, v) are patched into slot `leading`)
// or by a DPP wave shift (LEADER_LDS = false).  New x, v go back with per-lane dword stores.
// Needs P*C % 4 == 0 (16-byte records); otherwise k_move_w64 is used.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <int S, bool LEADER_LDS>
__global__ __launch_bounds__(256) void k_move_dma(const Dev d, const int tidx) {
  constexpr int TR = 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int tick = *d.tickA;
  const int C = d.C;
  const int stride = d.P * C;                   // floats per road record
  const int sub_floats = S * stride;            // multiple of 4
  float *buf = reinterpret_cast<float *>(smem) + (size_t)wv * sub_floats;

  const int G = gridDim.x;
  const int lb = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const long total = (long)d.E * d.R;
  const long tiles = (total + TR - 1) / TR;
  const long nw = (long)G * 4;
  const long chunk = (tiles + nw - 1) / nw;
  const long t0 = ((long)lb * 4 + wv) * chunk;
  const long t1 = (t0 + chunk < tiles) ? t0 + chunk : tiles;
  const int tick_sp = (d.spawn_mode == TFX_SPAWN_PERIODIC) ? tick % d.spawn_period : 0;

  unsigned long long my_updates = 0;

  for (long tile = t0; tile < t1; ++tile) {
    const long base = tile * TR;
    // ================= phase M: lane j <-> road base + j ========================================
    const bool valid = base + lane < total;
    const int id = valid ? (int)(base + lane) : (int)(total - 1);
    const int env = id / d.R;
    const int e = id - env * d.R;
    const bool train = e < d.r;
    const int dir = train ? e / d.I : 0;
    const int dst = e - dir * d.I;
    const int ld = d.leading[id];
    int lc = d.lastcar[id];
    const int n_old = ring_count(ld, lc, C);
    int n_tot = n_old, ovf_sp = 0;
    float xL = INFINITY, xs0 = 0.0f;
    if (train) {
      int ph_new, el_new;
      light_update(d, env, dst, tick, tidx, ph_new, el_new);
      const int phase_e = (dir < 2) ? 1 : 0;
      if (phase_e == ph_new || el_new < d.yellow) {
        xL = d.length;
      } else {
        const int idn = env * d.R + d.nexts[e];
        if (d.lastcar[idn] != d.leading[idn]) xL = d.tailx[idn] + d.length;
      }
    }
    const int ej = d.entry_idx[e];
    if (ej >= 0 && valid) {
      int c;
      if (d.spawn_mode == TFX_SPAWN_COUNTS)
        c = d.spawn[(size_t)tidx * d.spawn_stride + (size_t)env * d.n_entry + ej];
      else if (d.spawn_mode == TFX_SPAWN_PERIODIC)
        c = (tick_sp == e % d.spawn_period) ? 1 : 0;
      else
        c = 0;
      if (c > 0) {
        float tail_x = d.tailx[id];
        for (int q = 0; q < c; ++q) {
          const int pos = wrap1(lc + 1, C);
          const float start = (lc != ld) ? (tail_x - d.car_l) - d.car_s0 : INFINITY;
          if (pos != ld) {
            const float xv = (start < 0.0f) ? start : 0.0f;
            if (n_tot == n_old) xs0 = xv;
            ++n_tot;
            lc = pos;
            tail_x = xv;
          } else {
            ++ovf_sp;
          }
        }
        if (n_tot != n_old) d.lastcar[id] = lc;
      }
    }
    const int pk = pack_desc(ld, lc, n_old, n_tot);
    const int xL_bits = __float_as_int(xL), xs0_bits = __float_as_int(xs0);

    // ================= phase C: sub-tiles of S roads through LDS =================================
    const long left = total - base;
    const int nroads = left < TR ? (int)left : TR;
    const int idb = (int)base;
    float *tx = d.state + (size_t)base * stride;  // wave-uniform tile base, 16-byte aligned
    int r_k = 0, r_w = 0, r_d = 0, r_t = 0;

    for (int j0 = 0; j0 < nroads; j0 += S) {
      const int ns = (nroads - j0 < S) ? nroads - j0 : S;
      // ---- LDS-DMA: ns*stride floats = n16 chunks of 16 B, 64 chunks per wave instruction
      {
        const float *gsrc = tx + (size_t)j0 * stride;
        const int n16 = (ns * stride) >> 2;
        for (int q0 = 0; q0 < n16; q0 += 64) {
          const int q = q0 + lane;
          if (q < n16)
            __builtin_amdgcn_global_load_lds((gptr_t *)(gsrc + (size_t)q * 4), (lptr_t *)(buf + (size_t)q0 * 4),
                                             16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
#pragma unroll
      for (int jj = 0; jj < S; ++jj) {
        if (jj < ns) {
          const int j = j0 + jj;
          const int pkj = __builtin_amdgcn_readlane(pk, j);
          const float xLj = __int_as_float(__builtin_amdgcn_readlane(xL_bits, j));
          const int ldj = pkj & 511, lcj = (pkj >> 9) & 511, n_oldj = (pkj >> 18) & 127;
          const int n_totj = (int)((unsigned)pkj >> 25);
          float *rb = buf + jj * stride;          // this road's record in LDS
          int slot = ldj + 1 + lane;
          slot = (slot >= C) ? slot - (C - 1) : slot;
          const bool is_live = lane < n_totj;
          float x, v, xl, vl;
          if (LEADER_LDS) {
            // patch the fake leader into its ring slot, then every car reads ring slot - 1
            if (lane == 0) {
              rb[ldj] = xLj;
              rb[C + ldj] = 0.0f;
            }
            if (n_totj != n_oldj) {  // cars spawned this tick are not in memory yet
              if (lane >= n_oldj && is_live) {
                float xv = __int_as_float(__builtin_amdgcn_readlane(xs0_bits, j));
                for (int q = n_oldj; q < lane; ++q) xv = (xv - d.car_l) - d.car_s0;
                rb[slot] = xv;
                rb[C + slot] = d.car_v;
              }
            }
            __builtin_amdgcn_wave_barrier();
            const int sl = is_live ? slot : 1;
            const int prev = (sl == 1) ? C - 1 : sl - 1;
            const int pv = (lane == 0) ? ldj : prev;
            x = rb[sl];
            v = rb[C + sl];
            xl = rb[pv];
            vl = rb[C + pv];
          } else {
            const int sl = (lane < n_oldj) ? slot : 1;
            x = rb[sl];
            v = rb[C + sl];
            if (n_totj != n_oldj) {
              if (lane >= n_oldj && is_live) {
                float xv = __int_as_float(__builtin_amdgcn_readlane(xs0_bits, j));
                for (int q = n_oldj; q < lane; ++q) xv = (xv - d.car_l) - d.car_s0;
                x = xv;
                v = d.car_v;
              }
            }
            xl = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(xLj), __float_as_int(x), 0x138, 0xf, 0xf, false));
            vl = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
          }
          const float ll = (lane == 0) ? 0.0f : d.car_l;

          const float t_gap = v * d.car_T;
          const float appr = v * (v - vl);
          const float s_star = d.car_s0 + np_max0(t_gap + appr / d.two_sab);
          const float sgap = (xl - x) - ll;
          const float q = v / d.car_v0;
          const float qd = pow4_cr(q);
          const float u = s_star / (sgap + d.eps);
          const float dv = d.car_a * ((1.0f - qd) - u * u);
          const float dvr = dv * d.rate;
          const float dx = d.rate * v + (0.5f * dvr) * d.rate;
          const float xn = x + (dx > 0.0f ? dx : 0.0f * dx);
          const float vn = np_max0(v + dvr);

          const int off = j * stride + slot;
          if (is_live) {
            tx[off] = xn;
            tx[off + C] = vn;
          }
          if (lane == 0) tx[j * stride + ldj] = xLj;  // the leader's x stays in its slot
          if (n_totj != n_oldj && d.P == 3) {
            if (lane >= n_oldj && is_live) tx[off + 2 * C] = (float)tick;
          }
          bool c_wait;
          if (ldj > lcj)
            c_wait = ((slot <= lcj) ? xn : vn) < d.thresh;
          else
            c_wait = vn < d.thresh;
          const unsigned long long m_pop = __ballot(is_live && (xn > d.length));
          const int n_wait = __popcll(__ballot(is_live && c_wait));
          const int n_det = __popcll(__ballot(is_live && (xn > d.near_end)));
          int kpop = (~m_pop == 0ull) ? 64 : __builtin_ctzll(~m_pop);
          int tail_bits = 0;
          if (n_totj > 0) tail_bits = __builtin_amdgcn_readlane(__float_as_int(xn), n_totj - 1);
          if (kpop > 0) {
            const bool far = is_live && (xn > d.length) && ((xn - d.length) > d.length);
            const bool slow = (kpop > KP) || (__ballot(far) != 0ull);
            if (lane < kpop && lane < KP) {
              float w = 0.0f;
              if (d.P == 3) w = (lane >= n_oldj) ? (float)tick : rb[2 * C + slot];
              float *pc = d.popcar + ((size_t)(idb + j) * KP + lane) * 3;
              pc[0] = xn;
              pc[1] = vn;
              pc[2] = w;
            }
            kpop |= slow ? (1 << 30) : 0;
          }
          const bool mine = lane == j;
          r_k = mine ? kpop : r_k;
          r_w = mine ? n_wait : r_w;
          r_d = mine ? n_det : r_d;
          r_t = mine ? tail_bits : r_t;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }

    // ================= phase W: lane j writes road j's results ===================================
    if (valid) {
      const int kpop = r_k & 0xffff;
      if (train) {
        int *ob = d.obs + (size_t)env * d.obs_len;
        if (n_tot > 0) {
          d.waiting[(size_t)env * d.r + e] += r_w;
          ob[d.r + e] = r_d;
        }
        ob[e] = kpop;
        if (kpop > 0) d.passed_dst[(size_t)env * d.I + dst] = 1;
      }
      d.rec[id] = make_int4(kpop, ovf_sp, r_t, n_tot);
      if ((r_k >> 30) & 1) d.env_flag[env] = tick + 1;
      my_updates += (unsigned long long)n_tot;
    }
  }

  for (int off = 32; off > 0; off >>= 1) my_updates += __shfl_down(my_updates, off);
  if (lane == 0 && my_updates) atomicAdd(d.veh, my_updates);
  if (blockIdx.x == 0 && threadIdx.x == 0) *d.tickB = tick;
}

// ------------------------------------------------------------------------------------------------
// Ring pop + handoff for destination road e, pull form.  Returns the overflow count of pushes
// into e.  Exact restatement of what the reference's sequential loop (advance_finished_cars
// :117-135) does to road e, given that (a) e's own pops are its first k_e cars and (b) the cars
// pushed into e are the k_p cars its predecessor p popped:
//   - the loop visits roads in ascending order, so p's pushes see leading[e] BEFORE e's own pops
//     when p < e and AFTER them when p > e (both the ring-full test and the empty-road test of
//     add_car :100-105 read leading[e]);
//   - successive pushes queue behind each other: start = x_tail - l - s0 of the previous push.
// ------------------------------------------------------------------------------------------------
__device__ int advance_road(const Dev &d, int env, int e) {
  const int C = d.C;
  const int id = env * d.R + e;
  const int ld = d.leading[id];
  int lc = d.lastcar[id];
  const int4 rc = d.rec[id];
  const int k_e = rc.x;
  float tail_x = __int_as_float(rc.z);
  const int ld_post = ring_adv(ld, k_e, C);
  float *xs = d.state + ((size_t)id * d.P) * C;
  float *vs = xs + C;
  float *ws = xs + 2 * C;
  const float xL = (k_e > 0) ? xs[ld] : 0.0f;  // read before a push can reuse the old leader slot

  int ovf = 0;
  const int p = d.pred[e];
  if (p >= 0) {
    const int idp = env * d.R + p;
    const int k_p = d.rec[idp].x;
    if (k_p > 0) {
      const int ld_seen = (p < e) ? ld : ld_post;
      const float *pc = d.popcar + (size_t)idp * KP * 3;
      for (int j = 0; j < k_p; ++j) {
        const float xc = pc[j * 3 + 0] - d.length;  // state[e,xi,newlead] -= length (:130)
        const int pos = wrap1(lc + 1, C);
        const float start = (lc != ld_seen) ? (tail_x - d.car_l) - d.car_s0 : INFINITY;
        if (pos != ld_seen) {
          const float xv = (start < xc) ? start : xc;
          xs[pos] = xv;
          vs[pos] = pc[j * 3 + 1];
          if (d.P == 3) ws[pos] = pc[j * 3 + 2];
          lc = pos;
          tail_x = xv;
        } else {
          ++ovf;
        }
      }
      d.lastcar[id] = lc;
    }
  }
  if (k_e > 0) {
    d.leading[id] = ld_post;
    xs[ld_post] = xL;  // state[e,:,newlead] = state[e,:,leading[e]] (:133)
  }
  d.tailx[id] = tail_x;
  return ovf;
}

// Literal single-thread advance for one env (taken when a road popped more than TFX_KP cars or a
// handed-off car could itself be popped again this tick).  Follows :117-157 line by line.
__device__ void advance_env_serial(const Dev &d, int env, int tick) {
  const int C = d.C;
  int *ob = d.obs + (size_t)env * d.obs_len;
  float *rew = d.rewards + (size_t)env * d.I;
  int overflowed = 0;
  for (int i = 0; i < d.I; ++i) rew[i] = 0.0f;
  for (int e = 0; e < d.R; ++e) {
    const int sp = d.rec[env * d.R + e].y;  // spawn overflows happened before move_cars
    if (sp > 0) {
      overflowed = 1;
      if (e < d.r)
        for (int j = 0; j < sp; ++j) rew[e % d.I] -= d.ovf_pen;
    }
  }
  for (int e = 0; e < d.r; ++e) ob[e] = 0;
  for (int e = 0; e < d.R; ++e) {
    const int id = env * d.R + e;
    float *xs = d.state + ((size_t)id * d.P) * C;
    float *vs = xs + C;
    float *ws = xs + 2 * C;
    int ld = d.leading[id];
    while (ld != d.lastcar[id] && xs[wrap1(ld + 1, C)] > d.length) {
      const int newlead = wrap1(ld + 1, C);
      const int nr = d.nexts[e];
      if (nr >= 0) {
        ob[e] += 1;
        d.passed_dst[(size_t)env * d.I + e % d.I] = 1;
        xs[newlead] -= d.length;
        const int idn = env * d.R + nr;
        float *xn = d.state + ((size_t)idn * d.P) * C;
        const int lcn = d.lastcar[idn], ldn = d.leading[idn];
        const int pos = wrap1(lcn + 1, C);
        const float start = (lcn != ldn) ? (xn[lcn] - d.car_l) - d.car_s0 : INFINITY;
        if (pos != ldn) {
          const float xc = xs[newlead];
          xn[pos] = (start < xc) ? start : xc;
          xn[C + pos] = vs[newlead];
          if (d.P == 3) xn[2 * C + pos] = ws[newlead];
          d.lastcar[idn] = pos;
        } else {
          if (nr < d.r) rew[nr % d.I] -= d.ovf_pen;
          overflowed = 1;
        }
      } else if (d.validate && d.n_trips) {
        const int t = d.n_trips[env];
        if (d.trip_times && t < d.trip_cap)
          d.trip_times[(size_t)env * d.trip_cap + t] = ((float)tick - ws[newlead]) / 2.0f;
        d.n_trips[env] = t + 1;
      }
      xs[newlead] = xs[ld];
      ld = newlead;
      d.leading[id] = ld;
    }
  }
  for (int e = 0; e < d.R; ++e) {
    const int id = env * d.R + e;
    const int lc = d.lastcar[id];
    d.tailx[id] = d.state[((size_t)id * d.P) * C + lc];
  }
  if (overflowed) d.done_tick[env] = tick + 1;
}

__global__ __launch_bounds__(256) void k_advance(const Dev d, const int tidx) {
  const int tick = *d.tickB;
  const int n_exit = d.R - d.r;
  const int per_env = d.I + n_exit;
  const long total = (long)d.E * per_env;
  for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
       gid += (long)gridDim.x * blockDim.x) {
    const int env = (int)(gid / per_env);
    const int s = (int)(gid - (long)env * per_env);
    const bool serial = d.env_flag[env] == tick + 1;
    if (serial && s == 0) advance_env_serial(d, env, tick);
    if (s < d.I) {
      // intersection s: its four incoming roads s, I+s, 2I+s, 3I+s (roadgraph.py:38-39)
      int ph_new, el_new;
      light_update(d, env, s, tick, tidx, ph_new, el_new);
      if (!serial) {
        int ovf = 0;
#pragma unroll
        for (int dir = 0; dir < 4; ++dir) {
          const int e = dir * d.I + s;
          ovf += advance_road(d, env, e) + d.rec[env * d.R + e].y;
        }
        // rewards[:] = 0 (:233) then -= OVERFLOW_PENALTY per dropped car (:110): exact in fp32
        float rw = 0.0f;
        for (int j = 0; j < ovf; ++j) rw -= d.ovf_pen;
        d.rewards[(size_t)env * d.I + s] = rw;
        if (ovf > 0) d.done_tick[env] = tick + 1;
      }
      int *ob = d.obs + (size_t)env * d.obs_len + 2 * d.r;
      ob[s] = ph_new;
      ob[d.I + s] = el_new;
    } else if (!serial) {
      const int e = d.r + (s - d.I);
      const int ovf = advance_road(d, env, e);
      if (ovf > 0) d.done_tick[env] = tick + 1;
      if (d.validate && d.n_trips && s == d.I) {
        // advance_hack :153-154: trip times of cars leaving the map, in road order
        int t = d.n_trips[env];
        for (int x = d.r; x < d.R; ++x) {
          const int idx = env * d.R + x;
          const int kx = d.rec[idx].x;
          for (int j = 0; j < kx; ++j) {
            if (d.trip_times && t < d.trip_cap)
              d.trip_times[(size_t)env * d.trip_cap + t] =
                  ((float)tick - d.popcar[((size_t)idx * KP + j) * 3 + 2]) / 2.0f;
            ++t;
          }
        }
        d.n_trips[env] = t;
      }
    }
    if (gid == 0) *d.tickA = tick + 1;
  }
}

// traffic_env.py:259-272
__global__ void k_reset(const Dev d, const int *phase_init) {
  const long total = (long)d.E * d.R;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total;
       id += (long)gridDim.x * blockDim.x) {
    const int env = (int)(id / d.R);
    const int e = (int)(id - (long)env * d.R);
    float *xs = d.state + ((size_t)id * d.P) * d.C;
    xs[1] = INFINITY;
    for (int p = 1; p < d.P; ++p) xs[p * d.C + 1] = 0.0f;
    d.leading[id] = 1;
    d.lastcar[id] = 1;
    d.tailx[id] = 0.0f;
    d.rec[id] = make_int4(0, 0, 0, 0);
    int *ob = d.obs + (size_t)env * d.obs_len;
    if (e < d.r) {
      ob[e] = 0;
      d.waiting[(size_t)env * d.r + e] = 0;
    }
    if (e < d.I) {
      ob[2 * d.r + e] = phase_init[(size_t)env * d.I + e];
      ob[2 * d.r + d.I + e] = 0;
      d.passed_dst[(size_t)env * d.I + e] = 0;
    }
    if (e == 0) {
      d.done_tick[env] = 0;
      d.env_flag[env] = 0;
      if (d.n_trips) d.n_trips[env] = 0;
    }
  }
}

__global__ void k_refresh(const Dev d) {
  const long total = (long)d.E * d.R;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total;
       id += (long)gridDim.x * blockDim.x) {
    const int lc = d.lastcar[id];
    d.tailx[id] = d.state[((size_t)id * d.P) * d.C + lc];
  }
}

// traffic_env.py:64-78
__global__ void k_remi(const Dev d) {
  const long total = (long)d.E * d.I;
  for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
       gid += (long)gridDim.x * blockDim.x) {
    const int env = (int)(gid / d.I);
    const int i = (int)(gid - (long)env * d.I);
    const int cur = d.obs[(size_t)env * d.obs_len + 2 * d.r + i];
    const bool pd = d.passed_dst[gid] != 0;
    float rw = 0.0f;
    for (int dir = 0; dir < 4; ++dir) {
      const int e = dir * d.I + i;
      const int phase_e = (dir < 2) ? 1 : 0;
      const bool green = phase_e != cur;
      int *wp = d.waiting + (size_t)env * d.r + e;
      const bool waiting = *wp > 0;
      if (waiting && !green && !pd) rw -= 0.5f;
      else if (pd && green && !waiting) rw += 0.5f;
      *wp = 0;
    }
    d.rewards[gid] = rw;
    d.passed_dst[gid] = 0;
  }
}

__global__ void k_cars_on_roads(const Dev d, int *out) {
  const long total = (long)d.E * d.R;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total;
       id += (long)gridDim.x * blockDim.x)
    out[id] = ring_count(d.leading[id], d.lastcar[id], d.C);
}

__global__ void k_done(const Dev d, uint8_t *out, int since_tick) {
  for (int env = blockIdx.x * blockDim.x + threadIdx.x; env < d.E; env += gridDim.x * blockDim.x)
    out[env] = d.done_tick[env] > since_tick ? 1 : 0;
}

}  // namespace

// ================================================================================================
// host side
// ================================================================================================
struct tfx_handle_s {
  tfx_config cfg;
  Dev d;
  bool bound = false;
  int n_cu = 256;
  int wpr = 1;
  int grid_move = 0;
  std::vector<int32_t> h_dest, h_phases, h_nexts, h_pred, h_entry, h_entry_idx;
  int *dev_tables = nullptr;  // nexts | pred | entry_idx
  void *dev_scratch = nullptr;
  int32_t action_per_tick = 0, spawn_per_tick = 0;
  // optional per-kernel timing with HIP events on the launch stream (tfx_profile)
  std::vector<hipEvent_t> ev;
  int ev_ticks = 0, ev_used = 0;
  bool prof = false;
  // TFX_MOVE_VARIANT (A/B runs): 0 k_move_w64<2,DPP> (default) | 2 w64<2,LDS> | 5 w64<4,DPP> |
  // 6 w64<1,DPP> | 3 k_move_tile<2> | 4 k_move_tile<4> | 1 k_move<1>
  int move_variant = 0;
  size_t dma_lds(int S) const { return (size_t)4 * S * d.P * d.C * sizeof(float); }
};

namespace {

// GridRoad tables (roadgraph.py:26-64), built row by row rather than per road.
void build_tables(tfx_handle_s *h) {
  const int m = h->cfg.m, n = h->cfg.n, v = m * n, r = 4 * v, R = r + 2 * m + 2 * n;
  h->h_dest.assign(R, -1);
  h->h_phases.assign(R, 0);
  h->h_nexts.assign(R, -1);
  h->h_pred.assign(R, -1);
  for (int dir = 0; dir < 4; ++dir)
    for (int row = 0; row < m; ++row)
      for (int col = 0; col < n; ++col) {
        const int li = row * n + col, e = dir * v + li;
        h->h_dest[e] = li;
        h->h_phases[e] = dir < 2 ? 1 : 0;
        int nx;
        switch (dir) {
          case 0: nx = col < n - 1 ? e + 1 : r + n + row; break;          // eastbound -> east exits
          case 1: nx = col > 0 ? e - 1 : r + 2 * n + m + row; break;      // westbound -> west exits
          case 2: nx = row < m - 1 ? e + n : r + n + m + col; break;      // -> exits after the last row
          default: nx = row > 0 ? e - n : r + col; break;                 // -> exits before row 0
        }
        h->h_nexts[e] = nx;
      }
  for (int e = 0; e < R; ++e)
    if (h->h_nexts[e] >= 0) h->h_pred[h->h_nexts[e]] = e;
  // generate_entrypoints (roadgraph.py:42-51): a set bit removes that side
  const uint32_t spec = h->cfg.entry_spec;
  h->h_entry.clear();
  if (!(spec & 1u)) for (int row = 0; row < m; ++row) h->h_entry.push_back(n * row);
  if (!((spec >> 1) & 1u)) for (int row = 1; row <= m; ++row) h->h_entry.push_back(v + n * row - 1);
  if (!((spec >> 2) & 1u)) for (int col = 0; col < n; ++col) h->h_entry.push_back(2 * v + col);
  if (!((spec >> 3) & 1u)) for (int col = 0; col < n; ++col) h->h_entry.push_back(3 * v + n * (m - 1) + col);
  h->h_entry_idx.assign(R, -1);
  for (size_t j = 0; j < h->h_entry.size(); ++j) h->h_entry_idx[h->h_entry[j]] = (int)j;
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int check_handle(tfx_handle h, bool need_bound) {
  if (!h) return fail(TFX_EINVAL, "null handle");
  if (need_bound && !h->bound) return fail(TFX_ESTATE, "tfx_bind_buffers has not been called");
  return TFX_OK;
}

// Grid of the move kernel: every block resident at once (occupancy query), a multiple of 8 so the
// XCD-contiguous chunking applies, never more blocks than there is work.
template <typename K>
int move_grid(tfx_handle h, K kernel, long work_items_per_block, size_t dyn_lds = 0) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, dyn_lds) != hipSuccess || per_cu < 1)
    per_cu = 4;
  if (per_cu > 8) per_cu = 8;
  const long total = (long)h->d.E * h->d.R;
  const long need = (total + work_items_per_block - 1) / work_items_per_block;
  long g = (long)h->n_cu * per_cu;
  if (g > need) g = need;
  if (g >= 8) g -= g % 8;
  return (int)(g < 1 ? 1 : g);
}

int launch_move(tfx_handle h, int tidx, hipStream_t st) {
  const Dev &d = h->d;
  if (h->grid_move == 0) {
    switch (h->wpr) {
      case 1:
        switch (h->move_variant) {
          case 7: h->grid_move = move_grid(h, k_move_dma<8, false>, 256, h->dma_lds(8)); break;
          case 8: h->grid_move = move_grid(h, k_move_dma<8, true>, 256, h->dma_lds(8)); break;
          case 9: h->grid_move = move_grid(h, k_move_dma<16, false>, 256, h->dma_lds(16)); break;
          case 10: h->grid_move = move_grid(h, k_move_dma<4, false>, 256, h->dma_lds(4)); break;
          case 0: h->grid_move = move_grid(h, k_move_w64<2, false>, 256); break;
          case 2: h->grid_move = move_grid(h, k_move_w64<2, true>, 256); break;
          case 5: h->grid_move = move_grid(h, k_move_w64<4, false>, 256); break;
          case 6: h->grid_move = move_grid(h, k_move_w64<1, false>, 256); break;
          case 3: h->grid_move = move_grid(h, k_move_tile<2>, 256); break;
          case 4: h->grid_move = move_grid(h, k_move_tile<4>, 256); break;
          default: h->grid_move = move_grid(h, k_move<1>, 4); break;
        }
        break;
      case 2: h->grid_move = move_grid(h, k_move<2>, 2); break;
      default: h->grid_move = move_grid(h, k_move<4>, 1); break;
    }
  }
  const dim3 gr(h->grid_move), bl(256);
  switch (h->wpr) {
    case 1:
      switch (h->move_variant) {
        case 7: hipLaunchKernelGGL((k_move_dma<8, false>), gr, bl, h->dma_lds(8), st, d, tidx); break;
        case 8: hipLaunchKernelGGL((k_move_dma<8, true>), gr, bl, h->dma_lds(8), st, d, tidx); break;
        case 9: hipLaunchKernelGGL((k_move_dma<16, false>), gr, bl, h->dma_lds(16), st, d, tidx); break;
        case 10: hipLaunchKernelGGL((k_move_dma<4, false>), gr, bl, h->dma_lds(4), st, d, tidx); break;
        case 0: hipLaunchKernelGGL((k_move_w64<2, false>), gr, bl, 0, st, d, tidx); break;
        case 2: hipLaunchKernelGGL((k_move_w64<2, true>), gr, bl, 0, st, d, tidx); break;
        case 5: hipLaunchKernelGGL((k_move_w64<4, false>), gr, bl, 0, st, d, tidx); break;
        case 6: hipLaunchKernelGGL((k_move_w64<1, false>), gr, bl, 0, st, d, tidx); break;
        case 3: hipLaunchKernelGGL(k_move_tile<2>, gr, bl, 0, st, d, tidx); break;
        case 4: hipLaunchKernelGGL(k_move_tile<4>, gr, bl, 0, st, d, tidx); break;
        default: hipLaunchKernelGGL(k_move<1>, gr, bl, 0, st, d, tidx); break;
      }
      break;
    case 2: hipLaunchKernelGGL(k_move<2>, dim3(h->grid_move), dim3(256), 0, st, d, tidx); break;
    default: hipLaunchKernelGGL(k_move<4>, dim3(h->grid_move), dim3(256), 0, st, d, tidx); break;
  }
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

int grid_for(long items, int n_cu) {
  long g = (items + 255) / 256;
  const long cap = (long)n_cu * 8;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

int launch_advance(tfx_handle h, int tidx, hipStream_t st) {
  const Dev &d = h->d;
  const long items = (long)d.E * (d.I + d.R - d.r);
  hipLaunchKernelGGL(k_advance, dim3(grid_for(items, h->n_cu)), dim3(256), 0, st, d, tidx);
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

}  // namespace

extern "C" {

int tfx_abi_version(void) { return TFX_ABI_VERSION; }
const char *tfx_last_error(void) { return g_err.c_str(); }

int tfx_create(const tfx_config *cfg, tfx_handle *out) {
  if (!cfg || !out) return fail(TFX_EINVAL, "null argument");
  if (cfg->m < 1 || cfg->n < 1) return fail(TFX_EINVAL, "grid must be at least 1x1");
  if (cfg->capacity < 3) return fail(TFX_EINVAL, "capacity must be >= 3 (slot 0 + fake leader + 1 car)");
  if (cfg->capacity - 2 > 256) return fail(TFX_EINVAL, "capacity-2 > 256 cars per road is not supported");
  if (cfg->n_envs < 1) return fail(TFX_EINVAL, "n_envs must be >= 1");
  if (cfg->planes != 2 && cfg->planes != 3) return fail(TFX_EINVAL, "planes must be 2 (x,v) or 3 (x,v,w)");
  if (cfg->validate && cfg->planes != 3) return fail(TFX_EINVAL, "validate mode needs planes = 3 (spawn tick w)");
  if (cfg->car_delta != 4.0f) return fail(TFX_EINVAL, "only delta = 4 (the reference's archetype) is supported");
  if (!(cfg->length > 0.0f) || !(cfg->rate > 0.0f)) return fail(TFX_EINVAL, "length and rate must be > 0");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev < 1) return fail(TFX_EDEVICE, "no HIP device");
  tfx_handle_s *h = new (std::nothrow) tfx_handle_s();
  if (!h) return fail(TFX_ENOMEM, "out of host memory");
  h->cfg = *cfg;
  build_tables(h);
  if (const char *mv = getenv("TFX_MOVE_VARIANT")) h->move_variant = atoi(mv);
  if (h->move_variant >= 7 && (cfg->planes * cfg->capacity) % 4 != 0) h->move_variant = 0;  // records not 16-B multiples
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
    h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

  Dev &d = h->d;
  memset(&d, 0, sizeof d);
  d.I = cfg->m * cfg->n;
  d.r = 4 * d.I;
  d.R = d.r + 2 * cfg->m + 2 * cfg->n;
  d.C = cfg->capacity;
  d.E = cfg->n_envs;
  d.P = cfg->planes;
  d.n_entry = (int)h->h_entry.size();
  d.obs_len = 2 * d.r + 2 * d.I;
  d.yellow = cfg->yellow_ticks;
  d.learn_switch = cfg->learn_switch;
  d.validate = cfg->validate;
  d.env_off = cfg->env_id_offset;
  d.length = cfg->length;
  d.rate = cfg->rate;
  d.car_v = cfg->car_v; d.car_l = cfg->car_l; d.car_a = cfg->car_a; d.car_v0 = cfg->car_v0;
  d.car_b = cfg->car_b; d.car_T = cfg->car_T; d.car_s0 = cfg->car_s0;
  d.two_sab = 2.0f * sqrtf(cfg->car_a * cfg->car_b);  // 2 * np.sqrt(a*b) (traffic_env.py:54)
  d.eps = cfg->eps;
  d.thresh = cfg->thresh;
  d.near_end = cfg->length - cfg->detect_dist;
  d.ovf_pen = cfg->overflow_penalty;
  if ((long)d.E * d.R > 0x7fffffffL / 4) { delete h; return fail(TFX_EINVAL, "E*R too large"); }

  const int cars = d.C - 2;
  h->wpr = cars <= 64 ? 1 : (cars <= 128 ? 2 : 4);
  h->grid_move = 0;  // sized at the first launch from the kernel's occupancy

  // tables
  const size_t R = (size_t)d.R;
  if (hipMalloc((void **)&h->dev_tables, 3 * R * sizeof(int)) != hipSuccess) {
    delete h;
    return fail(TFX_ENOMEM, "hipMalloc(tables) failed");
  }
  if (hipMemcpy(h->dev_tables, h->h_nexts.data(), R * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(h->dev_tables + R, h->h_pred.data(), R * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(h->dev_tables + 2 * R, h->h_entry_idx.data(), R * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(h->dev_tables);
    delete h;
    return fail(TFX_EDEVICE, "uploading the road tables failed");
  }
  d.nexts = h->dev_tables;
  d.pred = h->dev_tables + R;
  d.entry_idx = h->dev_tables + 2 * R;

  // scratch
  const size_t ER = (size_t)d.E * R;
  size_t off = 0;
  const size_t o_rec = off;   off = align_up(off + ER * sizeof(int4), 256);
  const size_t o_pop = off;   off = align_up(off + ER * KP * 3 * sizeof(float), 256);
  const size_t o_tail = off;  off = align_up(off + ER * sizeof(float), 256);
  const size_t o_flag = off;  off = align_up(off + (size_t)d.E * sizeof(int), 256);
  const size_t o_misc = off;  off = align_up(off + 64, 256);
  if (hipMalloc(&h->dev_scratch, off) != hipSuccess) {
    (void)hipFree(h->dev_tables);
    delete h;
    return fail(TFX_ENOMEM, "hipMalloc(scratch, %zu bytes) failed", off);
  }
  if (hipMemset(h->dev_scratch, 0, off) != hipSuccess) {
    (void)hipFree(h->dev_tables);
    (void)hipFree(h->dev_scratch);
    delete h;
    return fail(TFX_EDEVICE, "clearing the scratch failed");
  }
  char *base = (char *)h->dev_scratch;
  d.rec = (int4 *)(base + o_rec);
  d.popcar = (float *)(base + o_pop);
  d.tailx = (float *)(base + o_tail);
  d.env_flag = (int *)(base + o_flag);
  d.veh = (unsigned long long *)(base + o_misc);
  d.tickA = (int *)(base + o_misc + 16);
  d.tickB = (int *)(base + o_misc + 32);
  d.action_mode = TFX_ACTION_CYCLE;
  d.action_period = 20;
  d.spawn_mode = TFX_SPAWN_NONE;
  d.spawn_period = 8;
  *out = h;
  return TFX_OK;
}

int tfx_destroy(tfx_handle h) {
  if (!h) return TFX_OK;
  for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
  if (h->dev_tables) (void)hipFree(h->dev_tables);
  if (h->dev_scratch) (void)hipFree(h->dev_scratch);
  delete h;
  return TFX_OK;
}

int tfx_dims(tfx_handle h, int32_t *I, int32_t *r, int32_t *R, int32_t *n_entry) {
  if (int rc = check_handle(h, false)) return rc;
  if (I) *I = h->d.I;
  if (r) *r = h->d.r;
  if (R) *R = h->d.R;
  if (n_entry) *n_entry = h->d.n_entry;
  return TFX_OK;
}

int tfx_tables(tfx_handle h, int32_t *dest, int32_t *phases, int32_t *nexts, int32_t *entrypoints) {
  if (int rc = check_handle(h, false)) return rc;
  const size_t R = (size_t)h->d.R;
  if (dest) memcpy(dest, h->h_dest.data(), R * sizeof(int32_t));
  if (phases) memcpy(phases, h->h_phases.data(), R * sizeof(int32_t));
  if (nexts) memcpy(nexts, h->h_nexts.data(), R * sizeof(int32_t));
  if (entrypoints) memcpy(entrypoints, h->h_entry.data(), h->h_entry.size() * sizeof(int32_t));
  return TFX_OK;
}

int tfx_bind_buffers(tfx_handle h, const tfx_buffers *b) {
  if (int rc = check_handle(h, false)) return rc;
  if (!b) return fail(TFX_EINVAL, "null buffers");
  if (!b->state || !b->leading || !b->lastcar || !b->obs || !b->rewards || !b->waiting ||
      !b->passed_dst || !b->done_tick)
    return fail(TFX_EINVAL, "state, leading, lastcar, obs, rewards, waiting, passed_dst and done_tick are required");
  if (h->cfg.validate && (!b->n_trips || (b->trip_times && b->trip_cap < 1)))
    return fail(TFX_EINVAL, "validate mode needs n_trips (and trip_cap >= 1 with trip_times)");
  Dev &d = h->d;
  d.state = b->state; d.leading = b->leading; d.lastcar = b->lastcar; d.obs = b->obs;
  d.rewards = b->rewards; d.waiting = b->waiting; d.passed_dst = b->passed_dst;
  d.done_tick = b->done_tick; d.trip_times = b->trip_times; d.n_trips = b->n_trips;
  d.trip_cap = b->trip_cap;
  h->bound = true;
  return TFX_OK;
}

int tfx_reset(tfx_handle h, const int32_t *phase_init, void *stream) {
  if (int rc = check_handle(h, true)) return rc;
  if (!phase_init) return fail(TFX_EINVAL, "phase_init is required");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipMemsetAsync(h->d.tickA, 0, sizeof(int), st));
  HIPCHK(hipMemsetAsync(h->d.tickB, 0, sizeof(int), st));
  hipLaunchKernelGGL(k_reset, dim3(grid_for((long)h->d.E * h->d.R, h->n_cu)), dim3(256), 0, st, h->d, phase_init);
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

int tfx_refresh(tfx_handle h, void *stream) {
  if (int rc = check_handle(h, true)) return rc;
  hipLaunchKernelGGL(k_refresh, dim3(grid_for((long)h->d.E * h->d.R, h->n_cu)), dim3(256), 0,
                     (hipStream_t)stream, h->d);
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

int tfx_set_actions(tfx_handle h, int32_t mode, const int32_t *dev, int32_t period, int32_t per_tick) {
  if (int rc = check_handle(h, false)) return rc;
  Dev &d = h->d;
  if (mode == TFX_ACTION_CYCLE) {
    if (period < 1) return fail(TFX_EINVAL, "cycle period must be >= 1");
    d.action_period = period;
  } else if (mode == TFX_ACTION_BUFFER || mode == TFX_ACTION_BROADCAST) {
    if (!dev) return fail(TFX_EINVAL, "action buffer is null");
    d.action = dev;
    d.action_stride = per_tick ? (mode == TFX_ACTION_BUFFER ? (long)d.E * d.I : (long)d.I) : 0;
  } else {
    return fail(TFX_EINVAL, "unknown action mode %d", mode);
  }
  d.action_mode = mode;
  h->action_per_tick = per_tick;
  return TFX_OK;
}

int tfx_set_spawns(tfx_handle h, int32_t mode, const int32_t *dev, int32_t period, int32_t per_tick) {
  if (int rc = check_handle(h, false)) return rc;
  Dev &d = h->d;
  if (mode == TFX_SPAWN_PERIODIC) {
    if (period < 1) return fail(TFX_EINVAL, "spawn period must be >= 1");
    d.spawn_period = period;
  } else if (mode == TFX_SPAWN_COUNTS) {
    if (!dev) return fail(TFX_EINVAL, "spawn buffer is null");
    d.spawn = dev;
    d.spawn_stride = per_tick ? (long)d.E * d.n_entry : 0;
  } else if (mode != TFX_SPAWN_NONE) {
    return fail(TFX_EINVAL, "unknown spawn mode %d", mode);
  }
  d.spawn_mode = mode;
  h->spawn_per_tick = per_tick;
  return TFX_OK;
}

int tfx_step(tfx_handle h, int32_t n_ticks, void *stream) {
  if (int rc = check_handle(h, true)) return rc;
  if (n_ticks < 0) return fail(TFX_EINVAL, "n_ticks < 0");
  hipStream_t st = (hipStream_t)stream;
  for (int t = 0; t < n_ticks; ++t) {
    const bool timed = h->prof && h->ev_used < h->ev_ticks;
    hipEvent_t *e = timed ? &h->ev[(size_t)h->ev_used * 3] : nullptr;
    if (timed) HIPCHK(hipEventRecord(e[0], st));
    if (int rc = launch_move(h, t, st)) return rc;
    if (timed) HIPCHK(hipEventRecord(e[1], st));
    if (int rc = launch_advance(h, t, st)) return rc;
    if (timed) {
      HIPCHK(hipEventRecord(e[2], st));
      ++h->ev_used;
    }
  }
  return TFX_OK;
}

int tfx_move_cars(tfx_handle h, void *stream) {
  if (int rc = check_handle(h, true)) return rc;
  return launch_move(h, 0, (hipStream_t)stream);
}

int tfx_advance_finished_cars(tfx_handle h, void *stream) {
  if (int rc = check_handle(h, true)) return rc;
  return launch_advance(h, 0, (hipStream_t)stream);
}

int tfx_remi(tfx_handle h, void *stream) {
  if (int rc = check_handle(h, true)) return rc;
  hipLaunchKernelGGL(k_remi, dim3(grid_for((long)h->d.E * h->d.I, h->n_cu)), dim3(256), 0,
                     (hipStream_t)stream, h->d);
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

int tfx_cars_on_roads(tfx_handle h, int32_t *out, void *stream) {
  if (int rc = check_handle(h, true)) return rc;
  if (!out) return fail(TFX_EINVAL, "out is null");
  hipLaunchKernelGGL(k_cars_on_roads, dim3(grid_for((long)h->d.E * h->d.R, h->n_cu)), dim3(256), 0,
                     (hipStream_t)stream, h->d, out);
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

int tfx_done(tfx_handle h, uint8_t *out, int32_t since_tick, void *stream) {
  if (int rc = check_handle(h, true)) return rc;
  if (!out) return fail(TFX_EINVAL, "out is null");
  hipLaunchKernelGGL(k_done, dim3(grid_for(h->d.E, h->n_cu)), dim3(256), 0, (hipStream_t)stream, h->d,
                     out, since_tick);
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

int tfx_get_tick(tfx_handle h, int32_t *tick) {
  if (int rc = check_handle(h, false)) return rc;
  if (!tick) return fail(TFX_EINVAL, "tick is null");
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(tick, h->d.tickA, sizeof(int), hipMemcpyDeviceToHost));
  return TFX_OK;
}

int tfx_set_tick(tfx_handle h, int32_t tick) {
  if (int rc = check_handle(h, false)) return rc;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(h->d.tickA, &tick, sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->d.tickB, &tick, sizeof(int), hipMemcpyHostToDevice));
  // tick stamps taken under the old clock must not alias ticks of the new one
  HIPCHK(hipMemset(h->d.env_flag, 0, (size_t)h->d.E * sizeof(int)));
  if (h->bound) HIPCHK(hipMemset(h->d.done_tick, 0, (size_t)h->d.E * sizeof(int)));
  return TFX_OK;
}

int tfx_vehicle_updates(tfx_handle h, uint64_t *out, void *stream) {
  if (int rc = check_handle(h, false)) return rc;
  if (!out) return fail(TFX_EINVAL, "out is null");
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  unsigned long long v = 0;
  HIPCHK(hipMemcpy(&v, h->d.veh, sizeof v, hipMemcpyDeviceToHost));
  *out = (uint64_t)v;
  return TFX_OK;
}

int tfx_reset_counters(tfx_handle h, void *stream) {
  if (int rc = check_handle(h, false)) return rc;
  HIPCHK(hipMemsetAsync(h->d.veh, 0, sizeof(unsigned long long), (hipStream_t)stream));
  return TFX_OK;
}

int tfx_profile(tfx_handle h, int32_t max_ticks) {
  if (int rc = check_handle(h, false)) return rc;
  for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
  h->ev.clear();
  h->ev_ticks = 0;
  h->ev_used = 0;
  h->prof = max_ticks > 0;
  if (!h->prof) return TFX_OK;
  h->ev.resize((size_t)max_ticks * 3);
  for (hipEvent_t &e : h->ev) HIPCHK(hipEventCreate(&e));
  h->ev_ticks = max_ticks;
  return TFX_OK;
}

int tfx_profile_read(tfx_handle h, double *move_ms, double *advance_ms, int32_t *n_ticks) {
  if (int rc = check_handle(h, false)) return rc;
  double mv = 0.0, ad = 0.0;
  for (int i = 0; i < h->ev_used; ++i) {
    float a = 0.f, b = 0.f;
    HIPCHK(hipEventSynchronize(h->ev[(size_t)i * 3 + 2]));
    HIPCHK(hipEventElapsedTime(&a, h->ev[(size_t)i * 3], h->ev[(size_t)i * 3 + 1]));
    HIPCHK(hipEventElapsedTime(&b, h->ev[(size_t)i * 3 + 1], h->ev[(size_t)i * 3 + 2]));
    mv += a;
    ad += b;
  }
  if (move_ms) *move_ms = mv;
  if (advance_ms) *advance_ms = ad;
  if (n_ticks) *n_ticks = h->ev_used;
  h->ev_used = 0;
  return TFX_OK;
}

int tfx_launch_info(tfx_handle h, int32_t *grid, int32_t *block, int32_t *waves_per_road) {
  if (int rc = check_handle(h, false)) return rc;
  if (h->grid_move == 0) return fail(TFX_ESTATE, "no move kernel has been launched yet");
  if (grid) *grid = h->grid_move;
  if (block) *block = 256;
  if (waves_per_road) *waves_per_road = h->wpr;
  return TFX_OK;
}

}  // extern "C"
