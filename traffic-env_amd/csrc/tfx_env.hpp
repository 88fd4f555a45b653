// tfx_env.hpp - k_env: ALL the ticks of a call for one env per workgroup, the cars streamed from HBM, the ring words
// resident in LDS.
//
// Up to round 3 a pair of ticks was  k_move_tt (the cars, a wavefront per tile of 64 roads, tiles of all envs spread
// over the chip)  +  k_tail (everything per road, a workgroup per env), the env range in two halves on two streams so
// that one half's k_tail ran under the other half's pass.  What that left on the table (PMC, cfg2): the per-road words
// - leading, lastcar, tailx, the road records - make about 190 bytes per road and pair of HBM traffic between the two
// kernels (written by one, read back by the other: 0.8 GB of a pair's 4.2 GB), and k_tail's bytes cost the pass their
// time one for one.  None of those words ever leaves its env (envs share nothing, traffic_env.py:361-382; the handoff
// goes to the next road of the SAME env, :117-135).
//
// So: one workgroup owns an env for the whole call.  It loads the env's ring words into LDS once (28 bytes per road +
// the light words: 32 KB at cfg2, four workgroups per CU), then for every pair of ticks
//     pass       its wavefronts walk the env's tiles (move_tt_tile: the body of k_move_tt, unchanged)      | barrier
//     advance(t) | barrier | the deferred cars' tick t+1 (edge_tile) | barrier | advance(t+1)             | barrier
// with every per-road word read and written in LDS through a device block whose pointers are shifted onto the copies
// (the phase code is k_advance's and k_edge's own), and writes the words back once at the end of the call.  HBM sees
// the cars (16 B per car and pair), the pass's second record (rec2: 16 B per road, written and read back a few
// microseconds later) and the outputs (obs, waiting, rewards).  No launch boundaries inside a call, no second stream,
// no k_risk launches: inside an agent step the workgroup evaluates k_risk's bound for its env right before each pair
// and takes the pair one tick at a time when it fails; an env that overflowed stops on the spot (`if done: break`).
//
// The clock: every workgroup reads tickA at its start; k_tick_add moves it behind the launch (as for k_res).
#pragma once
#include "tfx_tail.hpp"

namespace tfx {

#ifndef ENV_WAVES
#define ENV_WAVES 6  // wavefronts of a workgroup (TFX_ENV_THREADS overrides at run time, up to ENV_MAX_THREADS lanes)
#endif
constexpr int ENV_MAX_THREADS = 512;

// bytes of LDS one env's words take (rec | leading | lastcar | tailx | (taila) | lights)
inline size_t env_lds_bytes(int R, int I, bool het) { return tail_lds_bytes(R, I, het); }

template <bool GREEDY = false, bool AGENT = false, bool W = false, bool HET = false>
__global__ __launch_bounds__(ENV_MAX_THREADS) TT_ATTR(W) void k_env(const Dev d, const int n_ticks) {
  static_assert(!HET || W, "heterogeneous cars carry their table row in the side word");
  extern __shared__ int4 s_dyn[];  // rec[R] | leading[R] | lastcar[R] | tailx[R] | (HET: taila[R]) | lights[2 I]
  __shared__ float s_arch[HET ? TFX_MAX_ARCH * ARCH_W : 1];
  if (HET) load_arch(d, s_arch);
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwv = blockDim.x >> 6;
  const int tick0 = *d.tickA;
  const int per_env = d.I + (d.R - d.r);
  const int R = d.R;
  int4 *const s_rec = s_dyn;
  int *const s_ld = reinterpret_cast<int *>(s_rec + R);
  int *const s_lc = s_ld + R;
  float *const s_tx = reinterpret_cast<float *>(s_lc + R);
  int *const s_ta = reinterpret_cast<int *>(s_tx + R);
  int *const s_lt = s_ta + (HET ? R : 0);

  unsigned long long my_updates = 0;
  for (int env = blockIdx.x; env < d.E; env += gridDim.x) {
    Dev dl = d;
    const size_t base = (size_t)env * R;
    for (int e = threadIdx.x; e < R; e += blockDim.x) {
      s_rec[e] = d.rec[base + e];
      s_ld[e] = d.leading[base + e];
      s_lc[e] = d.lastcar[base + e];
      s_tx[e] = d.tailx[base + e];
      if (HET) s_ta[e] = d.taila[base + e];
    }
    {
      const int *lt = d.lights + (size_t)env * d.lights_stride;
      for (int i = threadIdx.x; i < 2 * d.I; i += blockDim.x) s_lt[i] = lt[i];
    }
    // the phase code indexes its arrays with env * R + e: pointers that land on the LDS copies for THIS env
    dl.rec = s_rec - base;
    dl.leading = s_ld - base;
    dl.lastcar = s_lc - base;
    dl.tailx = s_tx - base;
    if (HET) dl.taila = s_ta - base;
    dl.lights = s_lt;
    dl.lights_stride = 0;
    dl.no_stamps = 1;  // (the workgroup decides about its env's pairs itself: k_risk's stamps are not consulted)
    __syncthreads();

    for (int t = 0; t < n_ticks;) {
      const int tick = tick0 + t;
      if (AGENT && env_frozen(d, env, tick)) break;  // (workgroup-uniform: overflowed earlier in this decision)
      const int sp = (d.spawn_mode == TFX_SPAWN_PERIODIC) ? tick % d.spawn_period : 0;
      bool two = t + 1 < n_ticks;
      if (AGENT && two) {
        // could tick t overflow a ring, or hand over more cars than a pair's bookkeeping carries (risk_lane)?  Then
        // the env takes both ticks one at a time: whether t + 1 runs at all depends on what t does
        bool risky = false;
        for (int g = wv; g < d.G; g += nwv) risky = risk_lane(dl, (long)env * d.G + g, env, lane, tick, sp, t) || risky;
        two = !__syncthreads_or(risky ? 1 : 0);
      }
      for (int g = wv; g < d.G; g += nwv)
        my_updates += (unsigned long long)move_tt_tile<true, AGENT, W, HET>(dl, (long)env * d.G + g, env, lane, tick, sp, t, two, s_arch);
      __syncthreads();
      for (int s = threadIdx.x; s < per_env; s += blockDim.x) advance_item<true, HET, GREEDY, W>(dl, env, s, tick, t);
      if (two) {
        const int sp1 = (d.spawn_mode == TFX_SPAWN_PERIODIC) ? (tick + 1) % d.spawn_period : 0;
        __syncthreads();
        for (int g = wv; g < d.G; g += nwv)
          my_updates += (unsigned long long)edge_tile<AGENT, W, HET>(dl, (long)env * d.G + g, env, lane, tick + 1, sp1, t + 1, s_arch);
        __syncthreads();
        for (int s = threadIdx.x; s < per_env; s += blockDim.x) advance_item<true, HET, GREEDY, W>(dl, env, s, tick + 1, t + 1);
      }
      __syncthreads();
      t += two ? 2 : 1;
    }

    for (int e = threadIdx.x; e < R; e += blockDim.x) {
      d.rec[base + e] = s_rec[e];
      d.leading[base + e] = s_ld[e];
      d.lastcar[base + e] = s_lc[e];
      d.tailx[base + e] = s_tx[e];
      if (HET) d.taila[base + e] = s_ta[e];
    }
    {
      int *lt = d.lights + (size_t)env * d.lights_stride;
      for (int i = threadIdx.x; i < 2 * d.I; i += blockDim.x) lt[i] = s_lt[i];
    }
    __syncthreads();  // (the next env's loads overwrite the copies)
  }

  for (int off = 32; off > 0; off >>= 1) my_updates += __shfl_down(my_updates, off);
  if (lane == 0 && my_updates) veh_add(d.veh, my_updates);
}

}  // namespace tfx
