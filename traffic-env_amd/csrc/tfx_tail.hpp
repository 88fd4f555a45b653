// tfx_tail.hpp - k_tail: everything of a pair of ticks that is per ROAD rather than per car, in one launch.
//
// A two-tick pass (tfx_move_tt.hpp) is followed by  k_advance(t)  k_edge(t+1)  k_advance(t+1): three
// lane-per-road launches that each stream the per-road words of every env through HBM (PMC at cfg2: 156 + 555 +
// 156 MB, 157 us of a 937 us pair).  All three only ever touch ONE env's words at a time - the handoff goes from a
// road to its successor in the same env (advance_finished_cars, traffic_env.py:117-135), the lights and the tails
// update_lights reads (:81-94) are the env's own - so a workgroup that owns an env can run the three back to back
// with workgroup barriers between them: the dependencies that forced three launches are inside the workgroup.
//
// Round 4: the env's ring words live in LDS for the whole launch.  Round 3's k_tail was pure latency (PMC: 83 %
// of its wave-cycles waiting, 6 % VALU): every phase walked chains of three or four dependent global loads
// (leading -> rec -> pred -> rec[pred] -> outbox ...) through an L2 the other half's car pass keeps busy.  Now one
// coalesced burst brings leading | lastcar | rec | tailx | the light words (28 B per road + 8 B per intersection: 32 KB
// at cfg2) into LDS, the three phases - and, inside agent steps, k_risk's bound for the NEXT pair - run on the LDS copies
// through a device block whose pointers are shifted onto them (the phase code is the same as k_advance's and k_edge's),
// and one burst writes them back.  What still goes to memory per phase is one level: the head car and the pass's second
// record (rec2) for the edge work, the outbox rows of roads that popped.  Envs too big for a workgroup's LDS (cfg4: 16 640
// roads) take the three launches instead (tail_usable).
//
// Ordering inside the workgroup: every phase reads words other lanes of the SAME workgroup wrote in the phase
// before (ring indices, tails, road records: LDS; outbox rows, cars handed over: global).  __syncthreads() is a
// workgroup-scope release / acquire; all wavefronts of a workgroup share their CU's vector L1, so nothing more is needed
// (no word crosses a workgroup, hence no agent-scope fence - the cost that sank round 2's cross-workgroup fusion).
//
// Clock: the pass left tick t in tickB; this kernel leaves t + 2 in tickA for the next move kernel (nobody reads
// tickA inside this launch, and tickB is not written, so workgroups that start late read the same t).
#pragma once
#include "tfx_advance.hpp"
#include "tfx_common.hpp"
#include "tfx_move_tt.hpp"

namespace tfx {

// k_tail's flags
enum { TAIL_RISK_NEXT = 1, TAIL_LAST = 2, TAIL_SYNC = 4 };

// bytes of LDS the staged form needs for one env
inline size_t tail_lds_bytes(int R, int I, bool het) {
  // (+ k_tail<AGENT>: the fake leaders' x, 4 bytes, and the bound's hint, 1 byte, per road)
  return (size_t)R * (sizeof(int4) + 4 * sizeof(int) + (het ? sizeof(int) : 0)) + (size_t)2 * I * sizeof(int) + 2 * (((size_t)R + 15) & ~(size_t)15);
}

// AGENT: inside an agent step (tfx_agent_step).  Envs that stand still are skipped by the phases themselves; envs k_risk
// marked for this pair (env_risk == t + 1: their first tick could overflow, so the pass left them alone) get BOTH ticks
// here, one at a time, from this same workgroup.  flags & TAIL_RISK_NEXT: another pair follows in
// the same decision - the last phase evaluates k_risk's bound for it (the state it needs is what advance(t + 1) has
// just left in LDS), so only the first pair of a decision pays a k_risk launch.
// flags & TAIL_LAST: the last pair of its call - the outputs only a caller can read are stored (edge_tile's full_out).
// flags & TAIL_SYNC: no pair follows in this call - leading, lastcar and hb are stored; otherwise (plain cars, outside
// agent steps) only the road state words the next pass reads (Dev::rsw: 4 bytes per road instead of 9).
// W: validate mode - the cars' side words travel along (edge_tile)
// HET (implies W): heterogeneous cars - the advance carries the cars' table rows, the edge work reads their parameters
// from an LDS copy of the table
// As many registers as a wavefront of the pass it runs beside (TT_ATTR: 80, with the side-word plane 96): its
// wavefronts then fit the slots the pass's leave.
template <bool GREEDY = false, bool AGENT = false, bool W = false, bool HET = false>
__global__ __launch_bounds__(256) TT_ATTR(W) void k_tail(const Dev d, const int tidx, const int flags) {
  static_assert(!HET || W, "heterogeneous cars carry their table row in the side word");
  extern __shared__ int4 s_dyn[];  // rec[R] | leading[R] | lastcar[R] | tailx[R] | (HET: taila[R]) | lights[2 I] | hb[R] bytes
  __shared__ float s_arch[HET ? TFX_MAX_ARCH * ARCH_W : 1];
  if (HET) load_arch(d, s_arch);
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwv = blockDim.x >> 6;
  const int tick = *d.tickB;  // first tick of the pair
  const int per_env = d.I + (d.R - d.r);
  const int sp0 = (d.spawn_mode == TFX_SPAWN_PERIODIC) ? (tick + 1) % d.spawn_period : 0;
  const int R = d.R;
  const int risk_next = flags & TAIL_RISK_NEXT;
  const bool full_out = (flags & TAIL_LAST) != 0;
  int4 *const s_rec = s_dyn;
  int *const s_ld = reinterpret_cast<int *>(s_rec + R);
  int *const s_lc = s_ld + R;
  float *const s_tx = reinterpret_cast<float *>(s_lc + R);
  int *const s_ta = reinterpret_cast<int *>(s_tx + R);
  int *const s_lt = s_ta + (HET ? R : 0);
  uint8_t *const s_hb = reinterpret_cast<uint8_t *>(s_lt + 2 * d.I);
  uint8_t *const s_hint = s_hb + ((R + 15) & ~15);                                  // AGENT
  float *const s_lx = reinterpret_cast<float *>(s_hint + ((R + 15) & ~15));        // AGENT

  unsigned long long my_updates = 0;
  for (int env = blockIdx.x; env < d.E; env += gridDim.x) {
    Dev dl = d;
    const size_t base = (size_t)env * R;
    // (workgroup-uniform) an env k_risk sorted out of this pair: its first tick could overflow; an env that stands still
    const bool sorted_out = AGENT && risk_word(d, env, tidx) == tick + 1;
    const bool frozen0 = AGENT && env_frozen(d, env, tick);
    // The pass's record carries the ring indices (crec_pack) - unless the pass skipped the env, or the cars are
    // heterogeneous (no room): then leading / lastcar, the tail cache and the row offsets are loaded.
    const bool packed = !HET && !sorted_out && !frozen0;
    {
      // (Tried: the loads of 2 - 4 roads per lane in flight at once.  The kernel must stay within the registers of ONE
      // wavefront of the pass - 80 - to take the slots the other half's pass frees one for one: unrolled four times it
      // needed 112 and the cfg2 tick went from 0.399 to 0.425 ms; within 80 registers no unrolling changed the tick.)
      for (int e = threadIdx.x; e < R; e += blockDim.x) {
        const int2 c = d.crec[base + e];  // the pass's record in its 8-byte form
        s_rec[e] = crec_expand<HET>(c, crec_ovf<HET>(c.x) ? d.ovf_cnt[base + e] : 0, d.C);
        if (packed) {  // the tail cache is rewritten by advance(t) before anyone reads it; the pass left no row offsets
          s_ld[e] = (c.x >> 9) & 511;
          s_lc[e] = (c.x >> 18) & 511;
          s_hb[e] = 0;
        } else {
          s_ld[e] = d.leading[base + e];
          s_lc[e] = d.lastcar[base + e];
          s_tx[e] = d.tailx[base + e];
          s_hb[e] = d.hb[base + e];
        }
        if (HET) s_ta[e] = d.taila[base + e];
      }
      const int *lt = d.lights + (size_t)env * d.lights_stride;
      for (int i = threadIdx.x; i < 2 * d.I; i += blockDim.x) s_lt[i] = lt[i];
      // the phase code indexes its arrays with env * R + e: pointers that land on the LDS copies for THIS env
      dl.rec = s_rec - base;
      dl.leading = s_ld - base;
      dl.lastcar = s_lc - base;
      dl.tailx = s_tx - base;
      if (HET) dl.taila = s_ta - base;
      dl.hb = s_hb - base;
      if (AGENT) {
        dl.leadx = s_lx - base;
        dl.riskhint = sorted_out ? nullptr : s_hint - base;  // (a sorted-out env runs no edge work: no hints)
      }
      dl.lights = s_lt;
      dl.lights_stride = 0;
      __syncthreads();
    }
    if (AGENT && sorted_out) {
      // The pass left this env's tiles alone.  Both of its ticks run here, one at a time - the one-tick pass over the
      // env's tiles by this workgroup's wavefronts, then its advance, twice.  An env that does overflow in tick t stands
      // still from there on (move_tt_tile and advance_item skip a frozen env: `if done: break`, traffic_test.py:55).
      // (Round 3 gave these envs the one-tick form in the pass and two restricted launches behind every k_tail - ten
      // launches per decision and half, all returning at once at the benchmark's density.)
      if (threadIdx.x == 0) atomicAdd(d.slow_pairs, 1ull);
      for (int u = 0; u < 2; ++u) {
        const int spu = (d.spawn_mode == TFX_SPAWN_PERIODIC) ? (tick + u) % d.spawn_period : 0;
        if (u) __syncthreads();
        for (int g = wv; g < d.G; g += nwv)
          my_updates += (unsigned long long)move_tt_tile<false, AGENT, W, HET>(dl, (long)env * d.G + g, env, lane, tick + u, spu, tidx + u, false, s_arch);
        __syncthreads();
        for (int s = threadIdx.x; s < per_env; s += blockDim.x) advance_item<true, HET, GREEDY, W>(dl, env, s, tick + u, tidx + u);
      }
    } else {
      for (int s = threadIdx.x; s < per_env; s += blockDim.x) advance_item<true, HET, GREEDY, W>(dl, env, s, tick, tidx);
      __syncthreads();
      for (int g = wv; g < d.G; g += nwv)
        my_updates += (unsigned long long)edge_tile<AGENT, W, HET>(dl, (long)env * d.G + g, env, lane, tick + 1, sp0, tidx + 1, s_arch, full_out);
      __syncthreads();
      for (int s = threadIdx.x; s < per_env; s += blockDim.x) advance_item<true, HET, GREEDY, W>(dl, env, s, tick + 1, tidx + 1);
    }
    if (AGENT && risk_next) {
      // k_risk's bound for the pair that follows, on the state advance(t + 1) has just left
      __syncthreads();
      const int t2 = tick + 2;
      if (!env_frozen(d, env, t2)) {
        const int sp2 = (d.spawn_mode == TFX_SPAWN_PERIODIC) ? t2 % d.spawn_period : 0;
        bool risky = false;
        for (int g = wv; g < d.G; g += nwv) risky = risk_lane(dl, (long)env * d.G + g, env, lane, t2, sp2, tidx + 2) || risky;
        if (__builtin_amdgcn_ballot_w64(risky) != 0ull && lane == 0) risk_word(d, env, tidx + 2) = t2 + 1;
      }
    }
    {
      __syncthreads();
      for (int e = threadIdx.x; e < R; e += blockDim.x) {
        // (the road records themselves are dead: the next pass writes new ones)
        if (AGENT || HET || (flags & TAIL_SYNC)) {
          d.hb[base + e] = s_hb[e];
          d.leading[base + e] = s_ld[e];
          d.lastcar[base + e] = s_lc[e];
        } else {
          d.rsw[base + e] = rsw_pack(s_ld[e], s_lc[e], s_hb[e]);
        }
        d.tailx[base + e] = s_tx[e];
        if (HET) d.taila[base + e] = s_ta[e];
      }
      int *lt = d.lights + (size_t)env * d.lights_stride;
      for (int i = threadIdx.x; i < 2 * d.I; i += blockDim.x) lt[i] = s_lt[i];
      // the fake leaders' x (tfx_export_ring reads them): when the decision ends, or the env does - it overflowed in
      // this pair and stands still from here on with what its last tick left
      if (AGENT && !frozen0 && (full_out || d.done_tick[env] > *d.agent_first))
        for (int e = threadIdx.x; e < R; e += blockDim.x) d.leadx[base + e] = s_lx[e];
      __syncthreads();  // (the next env's loads overwrite the copies)
    }
  }

  for (int off = 32; off > 0; off >>= 1) my_updates += __shfl_down(my_updates, off);
  if (lane == 0 && my_updates) veh_add(d.veh, my_updates);
  if (blockIdx.x == 0 && threadIdx.x == 0) *d.tickA = tick + 2;
}

}  // namespace tfx
