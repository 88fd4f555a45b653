// tfx_move_t.hpp - k_move_t: the move kernel on the TRANSPOSED car layout (tfx_config.layout = 1).
//
// Layout: roads are grouped in tiles of 64 consecutive roads; the k-th car behind the fake leader
// of road j of a tile sits at T[tile][k][j] (position-major, road-minor, (x, v) pairs).  One
// wavefront owns a tile and lane j walks road j from the head to the tail:
//   * iteration k loads T[tile][k][0..63]: 64 lanes x 8 B, one fully coalesced 512-byte row,
//     straight into registers - no LDS, no DMA staging, no ring-slot arithmetic;
//   * the leader of car k is car k-1 of the same road = the values this lane held one iteration
//     earlier (Jacobi: the OLD ones), so the follower-gap needs no cross-lane traffic at all;
//     the fake leader (update_lights :81-94) seeds the chain;
//   * waiting / detected counts and the pop prefix (advance_finished_cars :123: cars leave from the
//     head while x > length) are per-lane running values - no ballots, no scans;
//   * survivors are written back compacted (car k goes to position k - pops so far) with row-shaped
//     stores that coalesce like the loads; the first TFX_KP = 2 cars that left go to the road's
//     outbox column (two rows per tile).  A road that pops MORE than two cars in a tick (only
//     pathological states do) is left uncompacted for this tick - from the third pop on every car,
//     popped or not, stays in its own row - and its env takes the serial advance, which reads the
//     popped cars from the outbox / the head rows and compacts afterwards (tfx_advance_t.hpp);
//   * rows beyond a road's car count are neither read nor written, so only live cars move.
// Rows k+1..k+P are in flight while row k is computed (register prefetch).  Per 64 cars this is
// ~65 vector instructions against ~110 per (up to 64-car) road for the lane-per-car kernels, and
// the bytes moved are the live ones.  Ring indices (leading/lastcar), counters and every float are
// bit-identical to the ring-layout kernels; tfx_export_ring / tfx_import_ring convert.
#pragma once
#include "tfx_common.hpp"

namespace tfx {

typedef float f2v __attribute__((ext_vector_type(2)));

// index of position 0 of road e of env in a transposed array (T or the outbox); position k is 64 k
// pairs further
__device__ __forceinline__ size_t tcol(const Dev &d, int env, int e) {
  const int s = d.road_slot[e];
  return ((size_t)env * d.G + (size_t)(s >> 6)) * (size_t)d.trows * 64 + (size_t)(s & 63);
}
// index of row 0 of road e's outbox column (KP rows per tile)
__device__ __forceinline__ size_t ocol_of(const Dev &d, int env, int e) {
  const int s = d.road_slot[e];
  return ((size_t)env * d.G + (size_t)(s >> 6)) * (size_t)KP * 64 + (size_t)(s & 63);
}
__device__ __forceinline__ size_t tpos(const Dev &d, int id, int k) {
  const int env = id / d.R;
  return tcol(d, env, id - env * d.R) + (size_t)k * 64;
}

// W: the spawn-tick plane travels with the cars (validate mode, advance_hack's trip times :139-157)
// HET (implies W): heterogeneous cars - the side word is 8 * spawn tick + the row of the archetype table the car was
// spawned from (tfx_config.n_archetypes); a car's own row gives T, s0, a, b, v0, delta, its leader's row the length
// the gap subtracts (sim reads ld[li], traffic_env.py:55); the table sits in LDS
template <int P, int NT = 0, bool W = false, bool HET = false>
__global__ __launch_bounds__(256) void k_move_t(const Dev d, const int tidx) {
  static_assert(!HET || W, "heterogeneous cars carry their table row in the side word");
  __shared__ float s_arch[HET ? TFX_MAX_ARCH * ARCH_W : 1];
  if (HET) load_arch(d, s_arch);
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tick = *d.tickA;
  const int C = d.C;
  const long tiles = (long)d.E * d.G;
  const long nw = (long)gridDim.x * 4;
  const int tick_sp = (d.spawn_mode == TFX_SPAWN_PERIODIC) ? tick % d.spawn_period : 0;

  unsigned long long my_updates = 0;

  for (long tile = (long)blockIdx.x * 4 + wv; tile < tiles; tile += nw) {
    const int env = (int)(tile / d.G);  // a tile never straddles envs
    const int e_slot = d.slot_road[(int)(tile - (long)env * d.G) * 64 + lane];
    const bool valid = e_slot >= 0;
    const int e = valid ? e_slot : 0;
    const int id = env * d.R + e;
    const bool run = valid && !env_frozen(d, env, tick);
    const RoadPrep p = prep_road(d, id, env, e, tick, tick_sp, tidx, run, run);
    const int n_old = run ? p.n_old : 0;
    const int n_sp = run ? p.n_tot - p.n_old : 0;

    float2 *col = d.xv + ((size_t)tile * d.trows) * 64 + lane;     // T[k] of this road = col[k * 64]
    float2 *ocol = d.outb + ((size_t)tile * KP) * 64 + lane;       // outbox column of this road (KP rows)
    float *wcol = W ? d.w + ((size_t)tile * d.trows) * 64 + lane : nullptr;
    float *owcol = W ? d.outw + ((size_t)tile * KP) * 64 + lane : nullptr;

    // longest road of the tile (wave-uniform loop bound)
    int kmax = n_old;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int o = __shfl_xor(kmax, off, 64);
      kmax = o > kmax ? o : kmax;
    }
    kmax = __builtin_amdgcn_readfirstlane(kmax);

    // per-road running state
    float xprev = p.xL, vprev = 0.0f, llv = 0.0f;  // leader of the next car: starts as the fake one
    int kpop = 0, n_wait = 0, n_det = 0;
    bool open = true, far = false;
    int shift = 0;                   // survivors move up by this many rows: the pops so far, or 0 once
                                     //   the road popped more than TFX_KP cars and stays uncompacted
    float tail_x = 0.0f;
    int last_a = 0;  // HET: table row of the last car processed (the road's tail after the move)
    // wrapped ring: x, not v, is tested on slots 1..lastcar (:210) = the cars from index kq on
    const int kq = (p.ld > p.lc) ? C - 1 - p.ld : 0x7fffffff;

    // One car of this lane's road.  Runs under `if (active)`: the EXEC mask keeps the running state
    // of lanes whose road is shorter than the tile's longest untouched.
    // NT bit 0: non-temporal loads, bit 1: non-temporal stores (every byte is touched once per tick)
    auto ld2 = [&](const float2 *ptr) {
      if (NT & 1) {
        const f2v t = __builtin_nontemporal_load(reinterpret_cast<const f2v *>(ptr));
        return make_float2(t.x, t.y);
      }
      return *ptr;
    };
    auto st2 = [&](float2 *ptr, float a, float b) {
      if (NT & 2) {
        f2v t;
        t.x = a;
        t.y = b;
        __builtin_nontemporal_store(t, reinterpret_cast<f2v *>(ptr));
      } else {
        *ptr = make_float2(a, b);
      }
    };
    auto step = [&](int k, float x, float v, float wv) {
      float xn, vn;
      if (HET) {
        last_a = side_arch(wv);
        const float *me = s_arch + last_a * ARCH_W;
        idm_step_het(d, me, x, v, xprev, vprev, llv, xn, vn);
        llv = me[AR_L];
      } else {
        const bool off_domain = __builtin_amdgcn_ballot_w64(!idm_fast_domain(v)) != 0ull;
        if (d.fastdiv && !off_domain) idm_step_fast(d, x, v, xprev, vprev, llv, xn, vn);
        else idm_step(d, x, v, xprev, vprev, llv, xn, vn);
        llv = d.car_l;
      }
      xprev = x;  // OLD state leads the next car (Jacobi)
      vprev = v;
      const bool pop = open && (xn > d.length);  // the while loop of :123
      open = pop;
      if (pop) {
        if (kpop < KP) {
          ocol[(size_t)kpop * 64] = make_float2(xn, vn);
          if (W) owcol[(size_t)kpop * 64] = wv;
          ++shift;
        } else {  // third pop: no survivor has been written yet - from here on every car stays in its row
          shift = 0;
          st2(&col[(size_t)k * 64], xn, vn);
          if (W) wcol[(size_t)k * 64] = wv;
        }
        far = far || ((xn - d.length) > d.length);
      } else {
        st2(&col[(size_t)(k - shift) * 64], xn, vn);
        if (W) wcol[(size_t)(k - shift) * 64] = wv;
      }
      kpop += pop ? 1 : 0;
      const float wq = (k >= kq) ? xn : vn;
      n_wait += (wq < d.thresh) ? 1 : 0;
      n_det += (xn > d.near_end) ? 1 : 0;
      tail_x = xn;
    };

    // ---- cars in memory: rows 0 .. kmax-1, P rows in flight ------------------------------------
    float2 pf[P];
    float pfw[P];
#pragma unroll
    for (int u = 0; u < P; ++u) {
      pf[u] = (u < n_old) ? ld2(&col[(size_t)u * 64]) : make_float2(0.0f, 0.0f);
      pfw[u] = (W && u < n_old) ? wcol[(size_t)u * 64] : 0.0f;
    }
    {
      for (int k0 = 0; k0 < kmax; k0 += P) {
#pragma unroll
        for (int u = 0; u < P; ++u) {
          const int k = k0 + u;
          if (k < kmax) {
            const float2 cur = pf[u];
            const float curw = pfw[u];
            if (k + P < kmax) {
              pf[u] = (k + P < n_old) ? ld2(&col[(size_t)(k + P) * 64]) : make_float2(0.0f, 0.0f);
              pfw[u] = (W && k + P < n_old) ? wcol[(size_t)(k + P) * 64] : 0.0f;
            }
            if (k < n_old) step(k, cur.x, cur.y, curw);
          }
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
        // OWN length and minimum gap - and brings the row add_new_cars drew for it (:164)
        int lc = ring_adv(p.ld, n_old, C);
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
            const float start = (lc != p.ld) ? (tx - s_arch[ta * ARCH_W + AR_L]) - s_arch[ta * ARCH_W + AR_S0] : INFINITY;
            const float xs = (start < 0.0f) ? start : 0.0f;
            step(n_old + s, xs, s_arch[row * ARCH_W + AR_V], side_pack(tick, row));
            lc = wrap1(lc + 1, C);
            tx = xs;
            ta = row;
          }
      } else {
        for (int s = 0; s < smax; ++s)
          if (s < n_sp) step(n_old + s, spawned_x(d, p.xs0, s), d.car_v, (float)tick);  // w = spawn tick
      }
    }

    // ---- phase W -------------------------------------------------------------------------------
    if (run) {
      const int n_tot = p.n_tot;
      if (e < d.r) {
        int *ob = d.obs + (size_t)env * d.obs_len;
        if (n_tot > 0) {
          d.waiting[(size_t)env * d.r + e] += n_wait;
          ob[d.r + e] = n_det;
        }
        ob[e] = (d.agent_mode && tidx > 0) ? ob[e] + kpop : kpop;
        if (kpop > 0) d.passed_dst[(size_t)env * d.I + e % d.I] = 1;
      }
      d.rec[id] = make_int4(rec_pack(kpop, p.ld, C), rec_y(p.ovf_sp, kpop > KP), __float_as_int(tail_x),
                            n_tot | (HET ? last_a << 16 : 0));
      if (far || kpop > KP) d.env_flag[env] = tick + 1;
      d.leadx[id] = p.xL;
      my_updates += (unsigned long long)n_tot;
    }
  }

  for (int off = 32; off > 0; off >>= 1) my_updates += __shfl_down(my_updates, off);
  if (lane == 0 && my_updates) veh_add(d.veh, my_updates);
  if (blockIdx.x == 0 && threadIdx.x == 0) *d.tickB = tick;
}

}  // namespace tfx
