// tfx_grid.hpp - k_grid: every tick of a tfx_step call in ONE launch for handles whose envs are too big for k_res but
// whose tiles all fit the chip at once (BASELINE config 5: one 64x64 grid with 128-car roads - 260 tiles).
//
// Tick by tick such a handle is bound by its launches: two kernels per tick, each as long as its fill, a chain of
// dependent loads and its drain (profiles/r04_cfg4_closed_loop_trace.txt: 17.8 + 7.3 us per tick for ~2 MB of cars).
// Here a workgroup OWNS a tile for the whole call: the tile's rows (64 roads x trows cars, 8 B each: 66.5 KB at cfg4) are
// loaded into LDS once, every tick's move (move_ts_tile, the body of k_move_ts: S wavefronts share the tile's walk) and
// handoff (advance_road_t, one lane per road, appending to the road's own column) run on the LDS copy, and the rows go
// back to HBM when the call ends.  What roads tell each other - the popped cars (outbox), the road records, ring
// indices, tail positions, light words - stays in global memory, and the workgroups meet at grid barriers:
//
//   move(t)  |B1|  handoff(t) + lights(t)  |B2|  [greedy decision for t+1  |B3|]  move(t+1) ...
//
// B1: a road pulls what its predecessor popped (another workgroup's outbox).  B2: the next move reads its successor's
// ring indices and tail (fake leader behind a red light) and the light words.  The greedy controller's decision
// (cars_on_roads of the four approaches AFTER the handoff, greedy.py:14-16) sits between two barriers of its own, on
// the ticks it is due.  An env that needs the literal serial handoff (advance_env_serial_t: a road popped more than
// TFX_KP cars, or a car travelled more than a road length) has its tiles written back, one lane runs the serial loop on
// global memory, and the tiles are read again - two more barriers, on that tick only.
//
// The grid barrier: a monotonic counter in device memory, one arrival per workgroup, agent-scope fences on both sides
// (the XCDs' L2s are written back / invalidated by them).  The launch is cooperative - HIP refuses it
// unless every workgroup is resident at once - and the wait is bounded all the same: a workgroup that waits longer than
// GRID_SPIN_LIMIT rounds raises the abort word (pinned host memory) and every workgroup leaves at its next barrier; the
// host reports it at the handle's next call.
// Results are those of k_move_ts + k_advance: the same device functions run on the same values.
#pragma once
#include "tfx_advance.hpp"
#include "tfx_advance_t.hpp"
#include "tfx_common.hpp"
#include "tfx_move_ts.hpp"

namespace tfx {

constexpr long GRID_SPIN_LIMIT = 1L << 21;  // rounds of s_sleep + one atomic load: a second or so

struct GridSync {
  unsigned *ctr;   // arrivals, monotonic over the launch (zeroed by the host before it)
  int *abort_word; // pinned host memory: raised by a workgroup whose wait ran out
  int *ovf;        // [2][E * I]: cars dropped at intersection i in tick t (plane t & 1), zeroed by the host
  unsigned n_wg;
  long long *prof;  // experiments: cycles per phase of workgroup 0 (s_memtime), or null
  int dbg;         // timing experiments only (TFX_GRID_DBG): 1 = no release fence, 2 = no acquire fence
};

// all threads of every workgroup of the launch; false: the launch is being abandoned.
// ONE wavefront per workgroup does the agent-scope work: the release fence (the XCD's L2 written back), the arrival, the
// wait and the acquire fence (L2 and the CU's vector cache invalidated - the cache all the workgroup's wavefronts read
// through).  The other wavefronts only make sure their own stores have left (vmcnt(0): a store is counted until L2 has
// it) and meet the first at workgroup barriers.  (Every wavefront fencing at agent scope - 16 x 260 write-backs and
// invalidations per barrier at cfg4 - was measured first: 150 us per barrier.)
__device__ __forceinline__ bool grid_barrier(const GridSync &g, unsigned &target, int &s_abort) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);  // (vmcnt / expcnt / lgkmcnt all zero)
  __syncthreads();
  if (threadIdx.x == 0) {
    target += g.n_wg;
    if (!(g.dbg & 1)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(g.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    int ab = 0;
    while (__hip_atomic_load(g.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      // (the abort word lives in host memory: looked at every 1024 rounds only)
      if (++spins > GRID_SPIN_LIMIT ||
          ((spins & 1023) == 0 && __hip_atomic_load(g.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)) {
        ab = 1;
        break;
      }
    }
    if (!(g.dbg & 2)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (ab) __hip_atomic_store(g.abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    s_abort = ab;
  }
  __syncthreads();
  return s_abort == 0;
}

// A workgroup of 64 * S lanes owns tiles blockIdx.x, blockIdx.x + gridDim.x, ... (at most `own` of them; at cfg4 the 260
// tiles of the env go to 256 workgroups - one per CU at this register count - and four of them, holding interior roads,
// also take one of the four tiles of exit roads, the lightest).  Dynamic LDS: the owned tiles' rows, float2 [own][trows][64].
template <int KS, int S, bool GREEDY>
__global__ __launch_bounds__(64 * S) void k_grid(const Dev d, const int n_ticks, const GridSync g) {
  extern __shared__ float2 s_rows[];
  __shared__ int s_wait[S][64], s_det[S][64], s_kpop[64];
  __shared__ float s_tail[64];
  __shared__ int s_abort;
  const int lane = threadIdx.x & 63;
  const int seg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long tiles = (long)d.E * d.G;
  const int n_row_words = d.trows * 64;
  const int n_own = (int)((tiles - blockIdx.x + gridDim.x - 1) / gridDim.x);
  const int tick0 = *d.tickA;
  unsigned long long my_updates = 0;
  unsigned target = 0;
  if (threadIdx.x == 0) s_abort = 0;

  // the owned tiles' rows, HBM <-> LDS
  auto rows_in = [&]() {
    for (int o = 0; o < n_own; ++o) {
      const float2 *src = d.xv + (size_t)(blockIdx.x + (long)o * gridDim.x) * n_row_words;
      for (int i = threadIdx.x; i < n_row_words; i += 64 * S) s_rows[(size_t)o * n_row_words + i] = src[i];
    }
  };
  auto rows_out = [&]() {
    for (int o = 0; o < n_own; ++o) {
      float2 *dst = d.xv + (size_t)(blockIdx.x + (long)o * gridDim.x) * n_row_words;
      for (int i = threadIdx.x; i < n_row_words; i += 64 * S) dst[i] = s_rows[(size_t)o * n_row_words + i];
    }
  };
  rows_in();
  __syncthreads();

  bool alive = true;
  int t = 0;
  long long pc[6] = {0, 0, 0, 0, 0, 0};
  long long c0 = __builtin_readcyclecounter();
  auto lap = [&](int i) { const long long c1 = __builtin_readcyclecounter(); pc[i] += c1 - c0; c0 = c1; };
  for (; t < n_ticks; ++t) {
    const int tick = tick0 + t;
    const int tick_sp = (d.spawn_mode == TFX_SPAWN_PERIODIC) ? tick % d.spawn_period : 0;
    for (int o = 0; o < n_own; ++o) {
      const long tile = blockIdx.x + (long)o * gridDim.x;
      // the device block whose car array is the LDS copy (every index the shared functions form for THIS tile's roads
      // lands in the tile's rows; they form no other)
      Dev dl = d;
      dl.xv = s_rows + (size_t)o * n_row_words - (size_t)tile * n_row_words;
      move_ts_tile<KS, false, S>(dl, tile, tick, tick_sp, t, lane, seg, s_wait, s_det, s_kpop, s_tail, my_updates);
    }
    lap(0);
    if (!(alive = grid_barrier(g, target, s_abort))) break;  // B1
    lap(1);

    // ---- handoff (advance_item's work, a lane per road of the tile; the intersection's own words by its first approach)
    bool any_serial = false;
    for (int q = 0; q < d.E; ++q) any_serial = any_serial || d.env_flag[q] == tick + 1;
    if (any_serial) {  // (every workgroup takes this branch or none: the flags are final since B1)
      rows_out();
      if (!(alive = grid_barrier(g, target, s_abort))) break;
      if (threadIdx.x == 0)
        for (int o = 0; o < n_own; ++o) {
          const long tile = blockIdx.x + (long)o * gridDim.x;
          const int env = (int)(tile / d.G);
          if (tile == (long)env * d.G && d.env_flag[env] == tick + 1) advance_env_serial_t<false, false>(d, env, tick, t);
        }
      if (!(alive = grid_barrier(g, target, s_abort))) break;
      rows_in();
      __syncthreads();
    }
    int *const ovf_now = g.ovf + (size_t)(t & 1) * d.E * d.I;
    int *const ovf_next = g.ovf + (size_t)((t + 1) & 1) * d.E * d.I;
    if (seg == 0) {
      for (int o = 0; o < n_own; ++o) {
        const long tile = blockIdx.x + (long)o * gridDim.x;
        const int env = (int)(tile / d.G);
        const int e = d.slot_road[(int)(tile - (long)env * d.G) * 64 + lane];
        if (e < 0) continue;
        const int id = env * d.R + e;
        if (d.env_flag[env] != tick + 1) {
          Dev dl = d;
          dl.xv = s_rows + (size_t)o * n_row_words - (size_t)tile * n_row_words;
          int ovf = advance_road_t<false, false>(dl, env, e);
          if (e < d.r) ovf += rec_ovf_sp(d.rec[id].y);
          if (ovf > 0) {
            d.done_tick[env] = tick + 1;
            if (e < d.r) atomicAdd(&ovf_now[(size_t)env * d.I + e % d.I], ovf);
          }
        }
        if (e < d.I) {  // the intersection's light words (TrafficEnv._step :225-232), committed once
          int ph_new, el_new;
          light_update(d, env, e, tick, t, ph_new, el_new);
          int *ob = d.lights + (size_t)env * d.lights_stride;
          ob[e] = ph_new;
          ob[d.I + e] = el_new;
          ovf_next[(size_t)env * d.I + e] = 0;  // (last added to a tick ago, read by nobody since)
        }
      }
    }
    lap(2);
    if (!(alive = grid_barrier(g, target, s_abort))) break;  // B2
    lap(3);

    if (GREEDY && d.greedy_spacing > 0 && (tick + 1) % d.greedy_spacing == 0) {
      if (seg == 0)
        for (int o = 0; o < n_own; ++o) {
          const long tile = blockIdx.x + (long)o * gridDim.x;
          const int env = (int)(tile / d.G);
          const int e = d.slot_road[(int)(tile - (long)env * d.G) * 64 + lane];
          if (e >= 0 && e < d.I) d.greedy_act[(size_t)env * d.I + e] = greedy_decide(d, env, e);
        }
      lap(4);
      if (!(alive = grid_barrier(g, target, s_abort))) break;  // B3
      lap(5);
    }
  }

  if (alive && n_ticks > 0 && seg == 0) {
    // rewards[:] = 0 (:233), then -= OVERFLOW_PENALTY per dropped car (:110) - of the call's LAST tick (a tfx_step call
    // outside agent steps accumulates nothing); an env on the serial path wrote its own
    const int tl = n_ticks - 1;
    for (int o = 0; o < n_own; ++o) {
      const long tile = blockIdx.x + (long)o * gridDim.x;
      const int env = (int)(tile / d.G);
      const int e = d.slot_road[(int)(tile - (long)env * d.G) * 64 + lane];
      if (e >= 0 && e < d.I && d.env_flag[env] != tick0 + tl + 1) {
        const int ovf = g.ovf[(size_t)(tl & 1) * d.E * d.I + (size_t)env * d.I + e];
        float rw = 0.0f;
        for (int j = 0; j < ovf; ++j) rw -= d.ovf_pen;
        d.rewards[(size_t)env * d.I + e] = rw;
      }
    }
  }
  __syncthreads();
  rows_out();
  if (seg == 0) {
    for (int off = 32; off > 0; off >>= 1) my_updates += __shfl_down(my_updates, off);
    if (lane == 0 && my_updates) veh_add(d.veh, my_updates);
  }
  if (g.prof && blockIdx.x == 0 && threadIdx.x == 0)
    for (int i = 0; i < 6; ++i) g.prof[i] += pc[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // (the ticks that ran: all of them unless the launch was abandoned, which the host reports)
    *d.tickA = tick0 + t;
    *d.tickB = tick0 + (t > 0 ? t - 1 : 0);
  }
}

}  // namespace tfx
