// tfx_tail.hpp - k_tail: everything of a pair of ticks that is per ROAD rather than per car, in one launch.
//
// A two-tick pass (tfx_move_tt.hpp) is followed by  k_advance(t)  k_edge(t+1)  k_advance(t+1): three
// lane-per-road launches that each stream the per-road words of every env through HBM (PMC at cfg2: 156 + 555 +
// 156 MB, 157 us of a 937 us pair).  All three only ever touch ONE env's words at a time - the handoff goes from a
// road to its successor in the same env (advance_finished_cars, traffic_env.py:117-135), the lights and the tails
// update_lights reads (:81-94) are the env's own - so a workgroup that owns an env can run the three back to back
// with workgroup barriers between them: the dependencies that forced three launches are inside the workgroup, and
// what k_edge writes for k_advance(t+1) (and k_advance(t) for k_edge) is found again in the L2 it was just
// written to.  One launch instead of three, and about half the HBM traffic.
//
// Ordering inside the workgroup: every phase reads words other lanes of the SAME workgroup wrote in the phase
// before (ring indices, tails, road records, outbox rows, light words).  __syncthreads() is a workgroup-scope
// release / acquire; all wavefronts of a workgroup share their CU's vector L1, so nothing more is needed (no word
// crosses a workgroup, hence no agent-scope fence - the cost that sank round 2's cross-workgroup fusion).
//
// Clock: the pass left tick t in tickB; this kernel leaves t + 2 in tickA for the next move kernel (nobody reads
// tickA inside this launch, and tickB is not written, so workgroups that start late read the same t).
#pragma once
#include "tfx_advance.hpp"
#include "tfx_common.hpp"
#include "tfx_move_tt.hpp"

namespace tfx {

// AGENT: inside an agent step (tfx_agent_step).  Envs that stand still are skipped by the phases themselves; envs k_risk
// marked for this pair (env_risk == t + 1: their first tick could overflow, so the pass took them through ONE tick) get
// the advance of tick t here and their whole second tick from the two restricted launches that follow
// (k_move_tt<false, true> with only_risky = 2, k_advance with only_risky = 1).
// W: validate mode - the cars' side words travel along (edge_tile)
// HET (implies W): heterogeneous cars - the advance carries the cars' table rows, the edge work reads their parameters
// from an LDS copy of the table
template <bool GREEDY = false, bool AGENT = false, bool W = false, bool HET = false>
__global__ __launch_bounds__(256) void k_tail(const Dev d, const int tidx) {
  static_assert(!HET || W, "heterogeneous cars carry their table row in the side word");
  __shared__ float s_arch[HET ? TFX_MAX_ARCH * ARCH_W : 1];
  if (HET) {
    if (threadIdx.x < TFX_MAX_ARCH * ARCH_W) s_arch[threadIdx.x] = d.arch_tab[threadIdx.x];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwv = blockDim.x >> 6;
  const int tick = *d.tickB;  // first tick of the pair
  const int per_env = d.I + (d.R - d.r);
  const int sp0 = (d.spawn_mode == TFX_SPAWN_PERIODIC) ? (tick + 1) % d.spawn_period : 0;

  unsigned long long my_updates = 0;
  for (int env = blockIdx.x; env < d.E; env += gridDim.x) {
    for (int s = threadIdx.x; s < per_env; s += blockDim.x) advance_item<true, HET, GREEDY, W>(d, env, s, tick, tidx);
    if (AGENT && d.env_risk[env] == tick + 1) continue;  // (workgroup-uniform; nobody waits at a barrier for it)
    __syncthreads();
    for (int g = wv; g < d.G; g += nwv)
      my_updates += (unsigned long long)edge_tile<AGENT, W, HET>(d, (long)env * d.G + g, env, lane, tick + 1, sp0, tidx + 1, s_arch);
    __syncthreads();
    for (int s = threadIdx.x; s < per_env; s += blockDim.x) advance_item<true, HET, GREEDY, W>(d, env, s, tick + 1, tidx + 1);
  }

  for (int off = 32; off > 0; off >>= 1) my_updates += __shfl_down(my_updates, off);
  if (lane == 0 && my_updates) veh_add(d.veh, my_updates);
  if (blockIdx.x == 0 && threadIdx.x == 0) *d.tickA = tick + 2;
}

}  // namespace tfx
