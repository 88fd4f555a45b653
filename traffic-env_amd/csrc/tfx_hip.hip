// tfx_hip.hip - MI355X (gfx950 / CDNA4) implementation of the IDM traffic-env tick behind the
// C ABI of include/tfx.h.  Written for wave64; no other target is supported.
//
// One tick (reference: gym_traffic/envs/traffic_env.py:224-248, TrafficEnv._step) is
//   * for envs that fit a compute unit's LDS: part of ONE launch per tfx_step / tfx_agent_step call -
//     k_res (tfx_resident.hpp), the cars resident on chip for all the ticks of the call;
//   * otherwise two kernels:
//       k_move_t / k_move_ts (transposed layout: tfx_move_t.hpp, tfx_move_ts.hpp)
//       or k_move_dma / k_move<WPR> (ring layout: tfx_move_dma.hpp, tfx_move_generic.hpp)
//                                                          lights, spawns, IDM, counts, compaction
//       k_advance (tfx_advance.hpp, tfx_advance_t.hpp)     ring pop + handoff, rewards, light words
// plus the cold kernels of tfx_misc.hpp.  This file is the host side: tables, scratch, launches.
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
#include "tfx_common.hpp"
#include "tfx_move_generic.hpp"
#include "tfx_move_dma.hpp"
#include "tfx_move_t.hpp"
#include "tfx_move_ts.hpp"
#include "tfx_move_tt.hpp"
#include "tfx_resident.hpp"
#include "tfx_advance.hpp"
#include "tfx_tail.hpp"
#include "tfx_misc.hpp"

using namespace tfx;

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

}  // namespace

struct tfx_handle_s {
  tfx_config cfg;
  Dev d;
  bool bound = false;
  int n_cu = 256;
  int wpr = 1;
  int grid_move = 0;
  int grid_tt[4] = {0, 0, 0, 0};  // k_move_tt<false>, <true>, <false, agent>, <true, agent>
  int grid_edge = 0;
  int grid_adv = 0;
  int grid_tail = 0;
  int tail = 1;               // k_tail after a two-tick pass (tfx_tail.hpp): TFX_TAIL=0 never, 2 at any batch size
  // calls of pairs run as TWO halves of the env range, the second on a stream of the handle's own: the latency-bound
  // per-road launch of one half (k_tail) runs under the other half's car pass (split_usable)
  int split = 1;              // TFX_SPLIT=0 never, 2 at any batch size
  hipStream_t split_stream = nullptr;
  hipEvent_t split_fork = nullptr, split_join = nullptr, split_stagger = nullptr;
  // which half of a split call is being enqueued (-1: none) and whether its first pass is still to come: the second
  // half's first pass waits for the first half's (from then on a half's pass runs under the other half's k_tail; left to
  // themselves both halves start their passes together and only fall into step a pair or two later)
  int split_half = -1;
  bool split_first = false;
  int *tick2 = nullptr;       // clock words of the second half (tickA, tickB), risk word
  bool het = false;           // heterogeneous cars (tfx_config.n_archetypes)
  float *dev_arch = nullptr;  // the archetype table on the device
  long long split_ticks = 0;  // ticks that ran split since tfx_create
  int pairs = 1;              // two-tick passes in tfx_step (tfx_move_tt.hpp): TFX_PAIRS=0 never, 2 at any size
  std::vector<int32_t> h_dest, h_phases, h_nexts, h_pred, h_entry, h_entry_idx, h_road_slot, h_slot_road;
  int *dev_tables = nullptr;  // nexts | pred | entry_idx | road_slot | slot_road
  int tiles_per_env = 0;      // G: 64-slot tiles one env occupies in the transposed layout
  // k_res (tfx_resident.hpp): whole envs resident in LDS for all the ticks of a call
  int res_epb = 0;            // envs per workgroup; 0 = the envs do not fit / disabled (TFX_RESIDENT=0)
  int res_lpr = 1;            // lanes per road (1 or 2)
  int res_threads = 0;
  size_t res_lds = 0;
  int res_min_ticks = 1;      // calls shorter than this take the per-tick kernels (TFX_RES_MIN_TICKS)
  void *dev_scratch = nullptr;
  int32_t action_per_tick = 0, spawn_per_tick = 0;
  // optional per-kernel timing with HIP events on the launch stream (tfx_profile)
  std::vector<hipEvent_t> ev;
  int ev_ticks = 0, ev_used = 0;
  std::vector<int> ev_weight;  // ticks the i-th timed entry covers (1, or the ticks of a fused launch)
  bool prof = false;
  long long fused_ticks = 0;   // ticks run by k_res since tfx_create
  long long pair_ticks = 0;    // ticks run as two-tick passes since tfx_create
  long long tail_ticks = 0;    // ... of which k_tail finished the pair (tfx_tail.hpp)
  long long ag_fused = 0, ag_pair = 0;  // what ONE replay of the captured agent-step graph adds to the two above
  const char *step_kernel = "";  // the kernel that moved the cars in the last tick (tfx_step_kernel)
  // TFX_MOVE_VARIANT selects the move kernel for A/B runs (see launch_move); 0 = best known
  int move_variant = 0;
  size_t move_lds = 0;
  unsigned long long div_mismatches = 0;  // result of the reciprocal-division self-test
  size_t n_tpairs = 0;                    // (x, v) pairs the xv buffer must hold in the transposed layout
  // fused agent step: the launch sequence of one step, captured once per (ticks, remi, inputs)
  hipGraph_t ag_graph = nullptr;
  hipGraphExec_t ag_exec = nullptr;
  hipStream_t ag_stream = nullptr;
  std::string ag_key;
  // bumped by every call that changes something a captured kernel argument was built from (bound
  // buffers, action / spawn rules, the Poisson stream): part of the graph key, so a stale graph is
  // never replayed even when a re-allocated buffer lands on the address the old one had
  unsigned long long input_gen = 0;
  bool use_graph = true;  // TFX_GRAPH=0 disables
  bool size_only = false;
  // on-device Poisson arrivals / greedy controller (own buffers)
  bool poisson = false, greedy = false;
  int greedy_spacing = 3;
  int poisson_rows = 1;        // ticks of arrival counts the Poisson buffer holds (tfx_step generates a call's worth up front)
  PoissonDev ps{};
  void *dev_ps = nullptr;      // counts | gap_left | draws | cdf
  int *dev_greedy = nullptr;   // [E][I] actions
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

// Storage slots of the transposed layout: road e of an env lives in slot road_slot[e].
// Roads of a kind behave alike - entry roads queue the arrivals, exit roads only drain - and a
// wavefront walks its tile as far as the tile's LONGEST road, so kinds are not mixed: interior
// train roads in id order (runs of consecutive ids: the per-road words still coalesce), then the
// entry roads, then the exit roads.  At cfg2 that is 15 + 1 + 1 tiles instead of ten tiles that
// each carry a few long entry roads (TFX_KINDS=0: plain id order).
void build_slots(tfx_handle_s *h) {
  const int R = (int)h->h_nexts.size(), r = 4 * h->cfg.m * h->cfg.n;
  const char *kv = getenv("TFX_KINDS");
  std::vector<int> order;
  if (!(kv && atoi(kv) == 0)) {
    for (int e = 0; e < r; ++e) if (h->h_pred[e] >= 0) order.push_back(e);
    for (int e = 0; e < r; ++e) if (h->h_pred[e] < 0) order.push_back(e);
    for (int e = r; e < R; ++e) order.push_back(e);
  } else {
    for (int e = 0; e < R; ++e) order.push_back(e);
  }
  h->tiles_per_env = (R + 63) / 64;
  h->h_slot_road.assign((size_t)h->tiles_per_env * 64, -1);
  for (int s = 0; s < R; ++s) h->h_slot_road[s] = order[s];
  h->h_road_slot.assign(R, -1);
  for (size_t s = 0; s < h->h_slot_road.size(); ++s)
    if (h->h_slot_road[s] >= 0) h->h_road_slot[h->h_slot_road[s]] = (int)s;
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
int move_grid(tfx_handle h, K kernel, long work_items_per_block, size_t dyn_lds = 0, int cap = 5) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, dyn_lds) != hipSuccess || per_cu < 1)
    per_cu = 4;
  // more resident waves than ~5 blocks per CU only adds concurrent DRAM streams: measured at cfg2
  // 3/4/5/6/7/8 blocks per CU -> 0.763/0.721/0.711/0.720/0.804/0.761 ms (k_move_t; k_move_tt takes 6, see there)
  if (per_cu > cap) per_cu = cap;
  if (const char *pc = getenv("TFX_MOVE_BLOCKS_PER_CU")) per_cu = atoi(pc) > 0 ? atoi(pc) : per_cu;
  const long total = h->d.layout == 1 ? (long)h->d.E * h->d.G * 64 : (long)h->d.E * h->d.R;
  const long need = (total + work_items_per_block - 1) / work_items_per_block;
  long g = (long)h->n_cu * per_cu;
  if (g > need) g = need;
  if (g >= 8) g -= g % 8;
  return (int)(g < 1 ? 1 : g);
}

// k_move_dma<CC, S, NBUF, UNR, LEADER_LDS>: size the grid on first use, then launch
template <int CC, int S, int NBUF, int UNR, bool LDSL, int LIVE = 0, int NP = 1>
int launch_dma(tfx_handle h, int tidx, hipStream_t st) {
  auto kern = k_move_dma<CC, S, NBUF, UNR, LDSL, LIVE, NP>;
  h->step_kernel = "k_move_dma";
  if (h->grid_move == 0) {
    h->move_lds = (size_t)4 * NBUF * S * h->d.C * sizeof(float2);
    if (h->move_lds > 64 * 1024)
      HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->move_lds));
    h->grid_move = move_grid(h, kern, 256, h->move_lds);
  }
  if (h->size_only) return TFX_OK;
  hipLaunchKernelGGL(kern, dim3(h->grid_move), dim3(256), h->move_lds, st, h->d, tidx);
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

template <int WPR>
int launch_generic(tfx_handle h, int tidx, hipStream_t st) {
  h->step_kernel = "k_move";
  if (h->grid_move == 0) h->grid_move = move_grid(h, k_move<WPR>, 256 / (64 * WPR));
  if (h->size_only) return TFX_OK;
  hipLaunchKernelGGL(k_move<WPR>, dim3(h->grid_move), dim3(256), 0, st, h->d, tidx);
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

// Transposed layout.  TFX_MOVE_VARIANT: 0 = automatic | 90 / 91 force / forbid the four-waves-per-tile kernel
// (tests run k_move_t at sizes the heuristics would give to k_move_ts)
int launch_move_t(tfx_handle h, int tidx, hipStream_t st) {
  const int pvar = h->move_variant;
  auto go = [&](auto kern) {
    if (h->grid_move == 0) h->grid_move = move_grid(h, kern, 256);
    if (h->size_only) return (int)TFX_OK;
    hipLaunchKernelGGL(kern, dim3(h->grid_move), dim3(256), 0, st, h->d, tidx);
    HIPCHK(hipGetLastError());
    return (int)TFX_OK;
  };
  // Launches too small to fill the chip with one wavefront per tile: four wavefronts per tile
  // (TFX_MOVE_VARIANT 90 forces it, 91 forbids it)
  const long tiles = (long)h->d.E * h->d.G;
  // measured (ms per launch, k_move_t -> k_move_ts): cfg2 x 16 envs (272 tiles) 0.055 -> 0.020, cfg4 x 1
  // (260) 0.075 -> 0.028, cfg1 x 256 (320) 0.023 -> 0.016, cfg4 x 4 (1040 tiles of 128 rows) 0.102 ->
  // 0.083; no gain at cfg2 x 64 (1088) and a loss at cfg1 x 1024 (1280): there the redundant road
  // prologues outweigh the shorter walks
  if (h->het) {  // heterogeneous cars: the one kernel that reads a car's parameters from its table row
    h->step_kernel = "k_move_t";
    return go(k_move_t<4, 3, true, true>);
  }
  const long split_below = (h->d.C - 2 > 64) ? (long)h->n_cu * 9 / 2 : (long)h->n_cu * 2;
  if (pvar == 90 || (tiles <= split_below && pvar == 0)) {
    auto gs = [&](auto kern, int threads = 256) {
      if (h->grid_move == 0) h->grid_move = (int)(tiles < (long)h->n_cu * 8 ? tiles : (long)h->n_cu * 8);
      if (h->size_only) return (int)TFX_OK;
      hipLaunchKernelGGL(kern, dim3(h->grid_move), dim3(threads), 0, st, h->d, tidx);
      HIPCHK(hipGetLastError());
      return (int)TFX_OK;
    };
    const int cap = h->d.C - 2;
    if (h->d.w) {
      if (cap <= 32) { h->step_kernel = "k_move_ts"; return gs(k_move_ts<8, true>); }
      if (cap <= 64) { h->step_kernel = "k_move_ts"; return gs(k_move_ts<16, true>); }
      if (cap <= 128) { h->step_kernel = "k_move_ts"; return gs(k_move_ts<32, true>); }
      { h->step_kernel = "k_move_ts"; return gs(k_move_ts<64, true>); }
    }
    if (cap <= 32) { h->step_kernel = "k_move_ts"; return gs(k_move_ts<8>); }
    if (cap <= 64) { h->step_kernel = "k_move_ts"; return gs(k_move_ts<16>); }
    // long roads on a launch of at most ~two tiles per CU: eight segments of 16 cars instead of four of 32
    // (cfg4 x 1 env closed loop: 38.4 -> 28.6 us per tick)
    if (cap <= 128 && cap > 64 && tiles <= (long)h->n_cu * 2) { h->step_kernel = "k_move_ts"; return gs(k_move_ts<16, false, 8>, 512); }
    if (cap <= 128) { h->step_kernel = "k_move_ts"; return gs(k_move_ts<32>); }
    { h->step_kernel = "k_move_ts"; return gs(k_move_ts<64>); }
  }
  if (h->d.w) { h->step_kernel = "k_move_t"; return go(k_move_t<4, 3, true>); }  // validate mode: the spawn-tick plane travels along
  // Cars that fit the 256 MiB Infinity Cache (+ L2) are found there again next tick: default caching
  // and 8 rows in flight.  Measured k_move_t<8> vs <4, nt> per launch: cfg2 x 128 envs (143 MB of cars)
  // 0.0365 vs 0.0392 ms, x 256 0.048 vs 0.054, x 384 0.068 vs 0.080, x 512 (286 MB) 0.086 vs 0.093,
  // cfg4 x 16 (272 MB) 0.096 vs 0.103, x 24 (409 MB) 0.134 vs 0.150; at cfg2 x 1024 (573 MB) it is the
  // other way round: 0.199 vs 0.173.  (12 or 16 rows in flight, or 8 resident blocks per CU: no better.)
  if (pvar == 0 && h->n_tpairs * sizeof(float2) <= (size_t)448 << 20) { h->step_kernel = "k_move_t"; return go(k_move_t<8>); }
  // beyond that every row is read once and written once per tick: non-temporal loads AND stores (0.82 -> 0.70 ms
  // at cfg2, and the following k_advance no longer waits for dirty lines: 0.057 -> 0.032 ms)
  { h->step_kernel = "k_move_t"; return go(k_move_t<4, 3>); }
}

// Ring layout.  TFX_MOVE_VARIANT: 0 = automatic | 1 generic k_move<1> | 26 k_move_dma with the capacity read at run time
int launch_move(tfx_handle h, int tidx, hipStream_t st) {
  if (h->d.layout == 1) return launch_move_t(h, tidx, st);
  const int C = h->d.C;
  const int v = h->move_variant;
  // cfg4: 128-car roads take two passes of a wavefront through the tiled kernel
  if (C == 130 && v != 1 && (long)h->d.E * h->d.R >= 64L * h->n_cu)
    return launch_dma<130, 8, 2, 2, false, 2, 2>(h, tidx, st);
  if (h->wpr == 2) return launch_generic<2>(h, tidx, st);
  if (h->wpr == 4) return launch_generic<4>(h, tidx, st);
  if ((C & 1) || v == 1) return launch_generic<1>(h, tidx, st);  // odd capacity: records not 16-B multiples
  // fewer roads than one 64-road tile per CU: the tiled kernel would leave most CUs idle and walk
  // its tile serially; one wavefront per road finishes sooner
  if (v == 0 && (long)h->d.E * h->d.R < 64L * h->n_cu) return launch_generic<1>(h, tidx, st);
  if (C == 34) return launch_dma<34, 8, 2, 4, false, 2>(h, tidx, st);   // cfg1
  if (C != 66 || v == 26) return launch_dma<0, 8, 1, 8, false>(h, tidx, st);  // capacity read at run time
  return launch_dma<66, 8, 2, 4, false, 2>(h, tidx, st);  // cfg2: best of the tuning runs
}

// sizes the move kernel's grid without launching (the occupancy queries must not run inside a
// stream capture)
int launch_move_probe(tfx_handle h) {
  h->size_only = true;
  const int rc = launch_move(h, 0, nullptr);
  h->size_only = false;
  return rc;
}

int grid_for(long items, int n_cu) {
  long g = (items + 255) / 256;
  const long cap = (long)n_cu * 8;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// The on-device Poisson stream for the next n_ticks ticks (rows of the count buffer); one workgroup per env, as
// many lanes as the burst is long (cfg4: thousands of cars per tick).
int launch_poisson(tfx_handle h, int n_ticks, hipStream_t st) {
  const Dev &d = h->d;
  const int threads = d.E <= 64 ? 1024 : (d.E <= 1024 ? 256 : 64);
  const int pg = d.E < h->n_cu * 16 ? d.E : h->n_cu * 16;
  hipLaunchKernelGGL(k_poisson, dim3(pg), dim3(threads), ((size_t)d.n_entry + 2) * sizeof(int), st, d, h->ps, n_ticks);
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

// Producers of the inputs of ONE tick when they are generated on the device tick by tick: the Poisson stream inside
// agent steps and single launches (tfx_step generates whole calls up front, see there).  The greedy controller's
// decisions are made by the advance of the tick before (advance_item); k_greedy runs once per call.
int launch_inputs(tfx_handle h, hipStream_t st) {
  if (h->poisson && h->d.spawn_stride == 0) return launch_poisson(h, 1, st);
  return TFX_OK;
}

int launch_greedy(tfx_handle h, hipStream_t st) {
  if (!h->greedy) return TFX_OK;
  hipLaunchKernelGGL(k_greedy, dim3(grid_for((long)h->d.E * h->d.I, h->n_cu)), dim3(256), 0, st, h->d,
                     h->dev_greedy, h->greedy_spacing);
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

// k_res: lanes per road and envs per workgroup (res_epb = 0: not applicable).  Limits: at most
// RES_MAX_THREADS lanes; the rings of the workgroup's envs in the LDS a workgroup may have (asked from
// the runtime with hipFuncSetAttribute: 160 KB per CU on gfx950).  Two lanes per road whenever one env
// fits that way (the walk of a road is the tick's critical path; TFX_RES_LPR=1 forces one).
template <int LPR, bool W>
bool res_try(tfx_handle h, int epb) {
  const Dev &d = h->d;
  const int threads = (LPR * epb * d.R + 63) / 64 * 64;
  if (threads > RES_MAX_THREADS) return false;
  size_t lds = res_lds_bytes(threads / LPR, d.C, epb, d.I, d.n_entry, W);
  if (lds > (size_t)160 * 1024) return false;
  {
    // Even placement: when every workgroup of the launch is resident at once the dispatcher may stack
    // seven of them on some CUs and one on others (measured: the same cfg1 x 1024 launch takes 10 or
    // 14.5 us per tick from run to run).  Asking for 1/b of a CU's LDS, b = workgroups per CU the launch
    // needs, leaves the dispatcher no such choice.
    const long grid = ((long)d.E + epb - 1) / epb;
    long per_cu = (grid + h->n_cu - 1) / h->n_cu;
    const long cap = (long)((size_t)160 * 1024 / lds);
    if (per_cu > cap) per_cu = cap;
    if (per_cu < 1) per_cu = 1;
    const size_t padded = ((size_t)160 * 1024 / (size_t)per_cu) & ~(size_t)255;
    if (grid >= h->n_cu && padded > lds) lds = padded;  // (a launch that cannot fill the chip has nothing to even out)
  }
  // The attribute belongs to the FUNCTION, not to the handle: handles with different LDS needs share it, so it is
  // only ever raised (a later, smaller handle must not pull it below what an earlier one launches with).
  static size_t granted = 64 * 1024;  // per instantiation <LPR, W>, process-wide
  if (lds > granted) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(static_cast<void (*)(const Dev, const ResArgs)>(k_res<LPR, W>)),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      return false;  // the runtime does not grant that much LDS
    }
    granted = lds;
  }
  h->res_lpr = LPR;
  h->res_epb = epb;
  h->res_threads = threads;
  h->res_lds = lds;
  return true;
}

template <bool W>
int res_configure(tfx_handle h) {
  const Dev &d = h->d;
  h->res_epb = 0;
  if (const char *rv = getenv("TFX_RESIDENT")) if (atoi(rv) == 0) return TFX_OK;
  if (const char *mt = getenv("TFX_RES_MIN_TICKS")) h->res_min_ticks = atoi(mt);
  int lpr_max = 2;
  if (const char *lv = getenv("TFX_RES_LPR")) lpr_max = atoi(lv) == 1 ? 1 : 2;
  const char *ev = getenv("TFX_RES_EPB");
  for (int lpr = lpr_max; lpr >= 1; --lpr) {
    auto fits = [&](int epb) { return lpr == 2 ? res_try<2, W>(h, epb) : res_try<1, W>(h, epb); };
    if (!fits(1)) continue;  // (leaves the one-env configuration in place)
    if (ev) {
      int want = atoi(ev) < 1 ? 1 : atoi(ev);
      if (want > d.E) want = d.E;
      while (want > 1 && !fits(want)) --want;
      return TFX_OK;
    }
    // two lanes per road: one env per workgroup measured best at every batch size (cfg1 x 1024: a
    // 10-tick call 100 us against 172 with two envs, x 4096: 403 against 533 with three) - fewer
    // wavefronts meet at each barrier.  One lane per road: 80-lane envs leave wavefronts half empty, so
    // pack envs: the smallest number of equal rounds over the chip, E / (CUs * b) for b = 1, 2, ...
    if (lpr == 2) return TFX_OK;
    for (int b = 1; b <= 64; ++b) {
      const int epb = (d.E + h->n_cu * b - 1) / (h->n_cu * b);
      if (epb <= 1 || fits(epb)) break;
    }
    return TFX_OK;
  }
  return TFX_OK;
}

// the resident kernel serves a call when the envs fit and trip times are not recorded (their order is
// the serial loop's)
bool res_usable(tfx_handle h, int n_ticks) {
  return h->res_epb > 0 && !h->d.validate && !h->het && n_ticks >= h->res_min_ticks;
}

int launch_res(tfx_handle h, int n_ticks, hipStream_t st, int tail = 0, int remi = 0, float *aobs = nullptr,
               float *areward = nullptr, uint8_t *adone = nullptr) {
  const Dev &d = h->d;
  ResArgs a;
  a.tail = tail;
  a.remi = remi;
  a.aobs = aobs;
  a.areward = areward;
  a.adone = adone;
  a.poisson = h->poisson ? 1 : 0;
  a.ps = h->ps;
  a.epb = h->res_epb;
  a.n_ticks = n_ticks;
  a.greedy_spacing = h->greedy ? h->greedy_spacing : 0;
  a.greedy_act = h->dev_greedy;
  const int grid = (d.E + h->res_epb - 1) / h->res_epb;
  a.own_clock = grid == 1 ? 1 : 0;
  h->step_kernel = "k_res";
  const dim3 g(grid), b(h->res_threads);
  if (h->res_lpr == 2) {
    if (d.w) hipLaunchKernelGGL((k_res<2, true>), g, b, h->res_lds, st, d, a);
    else hipLaunchKernelGGL((k_res<2, false>), g, b, h->res_lds, st, d, a);
  } else {
    if (d.w) hipLaunchKernelGGL((k_res<1, true>), g, b, h->res_lds, st, d, a);
    else hipLaunchKernelGGL((k_res<1, false>), g, b, h->res_lds, st, d, a);
  }
  HIPCHK(hipGetLastError());
  if (!a.own_clock) {  // every workgroup reads the clock at its start: it moves in a launch of its own
    hipLaunchKernelGGL(k_tick_add, dim3(1), dim3(1), 0, st, d, n_ticks);
    HIPCHK(hipGetLastError());
  }
  return TFX_OK;
}

int launch_advance(tfx_handle h, int tidx, hipStream_t st, int only_risky = 0) {
  const Dev &d = h->d;
  if (h->grid_adv == 0) {
    // no more blocks than are resident at once (k_advance<true> holds 5 per CU): with 8 per CU launched the
    // last three of every CU start when the first five have finished their whole grid-stride loop
    const long items = (long)d.E * (d.I + d.R - d.r);
    int per_cu = 0;
    const hipError_t qe = d.layout == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_advance<true>, 256, 0)
                                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_advance<false>, 256, 0);
    if (qe != hipSuccess || per_cu < 1) per_cu = 4;
    if (per_cu > 8) per_cu = 8;
    long g = (items + 255) / 256;
    if (g > (long)h->n_cu * per_cu) g = (long)h->n_cu * per_cu;
    h->grid_adv = (int)(g < 1 ? 1 : g);
  }
  if (h->size_only) return TFX_OK;
  const bool g = h->greedy;
  if (h->het) {
    if (g) hipLaunchKernelGGL((k_advance<true, true, true>), dim3(h->grid_adv), dim3(256), 0, st, d, tidx, only_risky);
    else hipLaunchKernelGGL((k_advance<true, true>), dim3(h->grid_adv), dim3(256), 0, st, d, tidx, only_risky);
  } else if (d.layout == 1) {
    if (g) hipLaunchKernelGGL((k_advance<true, false, true>), dim3(h->grid_adv), dim3(256), 0, st, d, tidx, only_risky);
    else hipLaunchKernelGGL(k_advance<true>, dim3(h->grid_adv), dim3(256), 0, st, d, tidx, only_risky);
  } else {
    if (g) hipLaunchKernelGGL((k_advance<false, false, true>), dim3(h->grid_adv), dim3(256), 0, st, d, tidx, only_risky);
    else hipLaunchKernelGGL(k_advance<false>, dim3(h->grid_adv), dim3(256), 0, st, d, tidx, only_risky);
  }
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

// Two ticks per pass over the cars (tfx_move_tt.hpp): for calls of two ticks or more on the transposed
// layout whose launches fill the chip, outside validate mode (the spawn-tick plane does not travel).  A handle
// that can use them (pairs_usable(h)) runs ALL its single ticks through k_move_tt<false>: the one-tick form that
// reads past the rows a pair may have left empty at the top of a column.
bool pairs_usable(tfx_handle h, int n_ticks = 2) {
  const Dev &d = h->d;
  // (a handle whose envs fit k_res never mixes the two: k_res loads its cars from row 0)
  if (!h->pairs || d.layout != 1 || d.w || n_ticks < 2 || h->move_variant != 0 || h->res_epb > 0) return false;
  // measured, vehicle-updates/s with / without: cfg2 x 16 envs (272 tiles) 1.8e10 / 2.5e10 and cfg4 x 1 (260) 2.6e10 /
  // 3.8e10 - there four wavefronts per tile (k_move_ts) finish sooner; cfg4 x 4 (1040) 9.0e10 / 6.6e10, cfg2 x 64
  // (1088) 9.1e10 / 7.1e10, cfg2 x 128 1.6e11 / 1.3e11, cfg4 x 8 2.1e11 / 1.8e11, cfg2 x 256 2.5e11 / 2.1e11
  const long tiles = (long)d.E * d.G;
  return h->pairs == 2 || tiles >= (long)h->n_cu * 4;
}

int edge_grid(tfx_handle h) {
  if (h->grid_edge == 0) {  // every block resident at once: a second, nearly empty round would double the time
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_edge<false>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    if (per_cu > 6) per_cu = 6;  // measured at cfg2, 4 / 5 / 6 / 7 blocks per CU: 0.112 / 0.102 / 0.097 / 0.118 ms
    const long tiles = (long)h->d.E * h->d.G;
    long g = (long)h->n_cu * per_cu;
    if (g > (tiles + 3) / 4) g = (tiles + 3) / 4;
    h->grid_edge = (int)(g < 1 ? 1 : g);
  }
  return h->grid_edge;
}

// AGENT: inside an agent step; only_risky: the second tick of the envs k_risk sorted out of a pair
template <bool TWO, bool AGENT = false>
int launch_move_tt(tfx_handle h, int tidx, hipStream_t st, int only_risky = 0) {
  // Grid: 10 workgroups per CU, 6 of them resident at once.  Measured at cfg2 (ms per pass alone on the chip
  // / vehicle-updates per second of the split call, same box): 6 workgroups per CU - every one resident for the whole
  // launch - 0.741 / 5.22-5.26e11; 10-12 per CU 0.706-0.719 / 5.26e11; 24 per CU 0.682 / 5.15e11; one tile per
  // wavefront (68 per CU) 0.681 / 5.12e11; another box 6 / 10 / 12 per CU: 0.749 / 0.716 / 0.722 and 5.13 / 5.14 /
  // 5.08e11.  A second, partial round of workgroups evens out the end of the launch; more rounds
  // cost the split call more than they give the launch.  (Grids of a whole number of workgroups per CU: an "exactly
  // balanced" 2902 instead of 3072 workgroups took 0.756.)
  int &resident = h->grid_tt[(TWO ? 1 : 0) + (AGENT ? 2 : 0)];
  if (resident == 0) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_move_tt<TWO, AGENT>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    if (per_cu > 6) per_cu = 6;
    resident = h->n_cu * per_cu;
  }
  h->step_kernel = "k_move_tt";
  if (h->size_only) return TFX_OK;
  long grid = (long)resident / 6 * 10;
  if (const char *pc = getenv("TFX_MOVE_BLOCKS_PER_CU")) grid = atoi(pc) > 0 ? (long)atoi(pc) * h->n_cu : grid;
  const long need = ((long)h->d.E * h->d.G + 3) / 4;
  if (grid > need) grid = need;
  if (grid >= 8) grid -= grid % 8;
  if (grid < 1) grid = 1;
  const bool stagger = TWO && h->split_first && h->split_half >= 0;
  if (stagger && h->split_half == 1) HIPCHK(hipStreamWaitEvent(st, h->split_stagger, 0));
  hipLaunchKernelGGL((k_move_tt<TWO, AGENT>), dim3((unsigned)grid), dim3(256), 0, st, h->d, tidx, only_risky);
  HIPCHK(hipGetLastError());
  if (stagger && h->split_half == 0) HIPCHK(hipEventRecord(h->split_stagger, st));
  if (stagger) h->split_first = false;
  return TFX_OK;
}

// The envs [lo, lo + n) of a handle as a Dev of their own: every per-env array starts at env lo, the global env
// id offset moves along (on-device rules are functions of the global id), the vehicle-update counter is shared.
Dev sub_dev(const tfx_handle_s *h, int lo, int n, int *clock) {
  Dev s = h->d;
  const Dev &d = h->d;
  const size_t R = (size_t)d.R, I = (size_t)d.I, r = (size_t)d.r, L = (size_t)lo;
  s.E = n;
  s.env_off = d.env_off + lo;
  const size_t tile_pairs = (size_t)d.G * (size_t)d.trows * 64, out_pairs = (size_t)d.G * KP * 64;
  s.xv = d.xv + L * (d.layout == 1 ? tile_pairs : R * d.C);
  if (d.w) s.w = d.w + L * (d.layout == 1 ? tile_pairs : R * d.C);
  s.leading = d.leading + L * R;
  s.lastcar = d.lastcar + L * R;
  s.obs = d.obs + L * (size_t)d.obs_len;
  s.rewards = d.rewards + L * I;
  s.waiting = d.waiting + L * r;
  s.passed_dst = d.passed_dst + L * I;
  s.done_tick = d.done_tick + L;
  if (d.trip_times) s.trip_times = d.trip_times + L * (size_t)d.trip_cap;
  if (d.n_trips) s.n_trips = d.n_trips + L;
  s.rec = d.rec + L * R;
  s.rec2 = d.rec2 + L * R;
  s.tailx = d.tailx + L * R;
  s.leadx = d.leadx + L * R;
  s.outb = d.outb + L * out_pairs;
  if (d.outw) s.outw = d.outw + L * out_pairs;
  s.env_flag = d.env_flag + L;
  s.env_risk = d.env_risk + L;
  if (d.action_mode == TFX_ACTION_BUFFER && d.action) s.action = d.action + L * I;
  if (d.greedy_act) s.greedy_act = d.greedy_act + L * I;
  if (d.spawn_mode == TFX_SPAWN_COUNTS && d.spawn) s.spawn = d.spawn + L * (size_t)d.n_entry;
  if (clock) {
    s.tickA = clock;
    s.tickB = clock + 1;
    s.risk_any = clock + 2;
  }
  return s;
}

// k_tail (tfx_tail.hpp) replaces k_advance(t) k_edge(t+1) k_advance(t+1) behind a two-tick pass: one workgroup per
// env.  Not when the arrivals of t+1 are produced by a launch between the two ticks (the Poisson stream tick by tick:
// agent steps; tfx_step generates them up front), and not below one env per CU (a handful of big envs - cfg4 - has
// too few workgroups to offer).  (The greedy controller decides inside the advance.)
bool tail_usable(tfx_handle h) {
  if (!h->tail || (h->poisson && h->d.spawn_stride == 0) || h->d.layout != 1 || h->d.w) return false;
  return h->tail == 2 || h->d.E >= h->n_cu;
}

// Two halves on two streams: where each half still runs pairs with k_tail behind them (one env per CU and half).
// Measured at cfg2, ms per tick, one range / two halves: 4096 envs 0.474 / 0.437, 2048 0.238 / 0.227, 1024
// 0.128 / 0.121, 512 0.0759 / 0.0705.  Not while per-kernel timing is on (tfx_profile times launches that own
// the chip).
bool split_usable(tfx_handle h, int n_ticks) {
  if (!h->split || h->prof || n_ticks < 2 || !pairs_usable(h, n_ticks) || !tail_usable(h)) return false;
  if (h->d.E < 2) return false;
  if (h->split == 2) return true;
  return h->d.E / 2 >= h->n_cu && (long)(h->d.E / 2) * h->d.G >= (long)h->n_cu * 4;
}

// (256 lanes per workgroup and as many workgroups as fit, measured at cfg2 against 512 / 1024 lanes and 2 / 3
// workgroups per CU: 0.088 ms per tick against 0.090-0.131 - the launch lives on wavefronts in flight)
int launch_tail(tfx_handle h, int tidx, hipStream_t st, bool agent = false) {
  if (h->grid_tail == 0) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (k_tail<false, false>), 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu > 5 && h->greedy) per_cu = 5;
    long g = (long)h->n_cu * per_cu;
    if (g > h->d.E) g = h->d.E;
    h->grid_tail = (int)(g < 1 ? 1 : g);
  }
  if (h->size_only) return TFX_OK;
  if (agent) {
    if (h->greedy) hipLaunchKernelGGL((k_tail<true, true>), dim3(h->grid_tail), dim3(256), 0, st, h->d, tidx);
    else hipLaunchKernelGGL((k_tail<false, true>), dim3(h->grid_tail), dim3(256), 0, st, h->d, tidx);
  } else {
    if (h->greedy) hipLaunchKernelGGL((k_tail<true, false>), dim3(h->grid_tail), dim3(256), 0, st, h->d, tidx);
    else hipLaunchKernelGGL((k_tail<false, false>), dim3(h->grid_tail), dim3(256), 0, st, h->d, tidx);
  }
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

template <bool AGENT>
int launch_edge(tfx_handle h, int tidx, hipStream_t st) {
  hipLaunchKernelGGL(k_edge<AGENT>, dim3(edge_grid(h)), dim3(256), 0, st, h->d, tidx);
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

int launch_risk(tfx_handle h, int tidx, hipStream_t st) {
  hipLaunchKernelGGL(k_risk, dim3(edge_grid(h)), dim3(256), 0, st, h->d, tidx);
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

}  // namespace

namespace {

// the second stream of a split call and the events that fork it from / join it to the caller's stream
int ensure_split(tfx_handle h) {
  if (h->split_stream) return TFX_OK;
  HIPCHK(hipStreamCreateWithFlags(&h->split_stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&h->split_fork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&h->split_join, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&h->split_stagger, hipEventDisableTiming));
  return TFX_OK;
}

// the launches of one agent step, in order, on `st`
// split: the ticks run as two halves of the env range, the second on the handle's own stream (as step_chunk does for
// tfx_step; launched eagerly - a batch big enough to split is not bound by its launches)
int agent_sequence(tfx_handle h, int n_ticks, int remi, float *aobs, float *areward, uint8_t *adone,
                   hipStream_t st, long long &n_fused, long long &n_pair, bool split = false) {
  Dev &d = h->d;
  n_fused = n_pair = 0;
  const int keep_mode = d.agent_mode, keep_acc = d.accum_rewards;
  if (res_usable(h, n_ticks)) {
    // every tick of the decision AND its tail (remi, observation, rewards, done flags) in one launch
    d.agent_mode = 1;
    d.accum_rewards = remi ? 0 : 1;
    const int rc = launch_res(h, n_ticks, st, 1, remi, aobs, areward, adone);
    d.agent_mode = keep_mode;
    d.accum_rewards = keep_acc;
    if (rc == TFX_OK) n_fused = n_ticks;
    return rc;
  }
  hipLaunchKernelGGL(k_agent_begin, dim3(1), dim3(1), 0, st, d, const_cast<int *>(d.agent_first));
  HIPCHK(hipGetLastError());
  if (int rc = launch_greedy(h, st)) return rc;
  d.agent_mode = 1;
  d.accum_rewards = remi ? 0 : 1;
  int rc = TFX_OK;
  const Dev whole = h->d;
  if (split) {
    h->size_only = true;  // (grids are sized for the whole range)
    (void)launch_move_tt<true, true>(h, 0, nullptr);
    (void)launch_move_tt<false, true>(h, 0, nullptr);
    (void)launch_tail(h, 0, nullptr, true);
    (void)launch_advance(h, 0, nullptr);
    h->size_only = false;
    (void)edge_grid(h);
    HIPCHK(hipEventRecord(h->split_fork, st));
    HIPCHK(hipStreamWaitEvent(h->split_stream, h->split_fork, 0));
    hipLaunchKernelGGL(k_clock_copy, dim3(1), dim3(1), 0, h->split_stream, whole.tickA, whole.tickB, h->tick2);
    HIPCHK(hipGetLastError());
  }
  hipStream_t user_st = st;
  for (int half = 0; half < (split ? 2 : 1) && rc == TFX_OK; ++half) {
    if (split) {
      const int n0 = whole.E / 2;
      h->d = half == 0 ? sub_dev(h, 0, n0, nullptr) : sub_dev(h, n0, whole.E - n0, h->tick2);
      st = half == 0 ? user_st : h->split_stream;
      h->split_half = half;
      h->split_first = true;
    }
    int t = 0;
    const bool tt = pairs_usable(h);
    if (tt) {
      // two-tick passes (tfx_move_tt.hpp); envs in which the first tick of a pair could overflow take the pair one
      // tick at a time (k_risk)
      for (; t + 1 < n_ticks && rc == TFX_OK; t += 2) {
        rc = launch_inputs(h, st);
        if (rc == TFX_OK) rc = launch_risk(h, t, st);
        if (rc == TFX_OK) rc = launch_move_tt<true, true>(h, t, st);
        if (rc == TFX_OK && tail_usable(h)) {
          // the rest of the pair in one launch; the envs k_risk sorted out get their second tick behind it
          rc = launch_tail(h, t, st, true);
          if (rc == TFX_OK) rc = launch_move_tt<false, true>(h, t + 1, st, 2);
          if (rc == TFX_OK) rc = launch_advance(h, t + 1, st, 1);
        } else {
          if (rc == TFX_OK) rc = launch_advance(h, t, st);
          if (rc == TFX_OK) rc = launch_inputs(h, st);
          if (rc == TFX_OK) rc = launch_edge<true>(h, t + 1, st);
          if (rc == TFX_OK) rc = launch_move_tt<false, true>(h, t + 1, st, 1);
          if (rc == TFX_OK) rc = launch_advance(h, t + 1, st);
        }
        if (rc == TFX_OK) n_pair += 2;
      }
    }
    for (; t < n_ticks && rc == TFX_OK; ++t) {
      rc = launch_inputs(h, st);
      if (rc == TFX_OK) rc = tt ? launch_move_tt<false, true>(h, t, st) : launch_move(h, t, st);
      if (rc == TFX_OK) rc = launch_advance(h, t, st);
    }
    if (split) h->d = whole;
  }
  h->split_half = -1;
  st = user_st;
  if (split) {
    n_pair /= 2;  // (both halves counted them)
    if (rc == TFX_OK) {
      HIPCHK(hipEventRecord(h->split_join, h->split_stream));
      HIPCHK(hipStreamWaitEvent(st, h->split_join, 0));
    }
  }
  d.agent_mode = keep_mode;
  d.accum_rewards = keep_acc;
  if (rc != TFX_OK) return rc;
  if (remi) {
    hipLaunchKernelGGL(k_remi, dim3(grid_for((long)d.E * d.I, h->n_cu)), dim3(256), 0, st, d);
    HIPCHK(hipGetLastError());
  }
  if (aobs) {
    hipLaunchKernelGGL(k_agent_obs, dim3(grid_for((long)d.E * (2 * d.r + d.I), h->n_cu)), dim3(256), 0, st, d, aobs);
    HIPCHK(hipGetLastError());
  }
  if (areward)
    HIPCHK(hipMemcpyAsync(areward, d.rewards, (size_t)d.E * d.I * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (adone) {
    hipLaunchKernelGGL(k_done_since, dim3(grid_for(d.E, h->n_cu)), dim3(256), 0, st, d, adone, d.agent_first);
    HIPCHK(hipGetLastError());
  }
  return TFX_OK;
}

}  // namespace

namespace {
// the per-tick kernels for n_ticks ticks of the envs h->d describes (the whole handle, or one half of it), on st
int step_range(tfx_handle h, int n_ticks, hipStream_t st) {
  int t = 0;
  const bool tt = pairs_usable(h);
  if (tt) {
    for (; t + 1 < n_ticks; t += 2) {
      const bool timed = h->prof && h->ev_used < h->ev_ticks;
      hipEvent_t *e = timed ? &h->ev[(size_t)h->ev_used * 3] : nullptr;
      if (int rc = launch_inputs(h, st)) return rc;
      if (timed) HIPCHK(hipEventRecord(e[0], st));
      if (int rc = launch_move_tt<true>(h, t, st)) return rc;
      if (timed) HIPCHK(hipEventRecord(e[1], st));
      if (tail_usable(h)) {
        if (int rc = launch_tail(h, t, st)) return rc;
        h->tail_ticks += 2;
      } else {
        if (int rc = launch_advance(h, t, st)) return rc;
        if (int rc = launch_inputs(h, st)) return rc;
        if (int rc = launch_edge<false>(h, t + 1, st)) return rc;
        if (int rc = launch_advance(h, t + 1, st)) return rc;
      }
      if (timed) {
        HIPCHK(hipEventRecord(e[2], st));
        h->ev_weight[h->ev_used] = 2;
        ++h->ev_used;
      }
      h->pair_ticks += 2;
    }
  }
  for (; t < n_ticks; ++t) {
    const bool timed = h->prof && h->ev_used < h->ev_ticks;
    hipEvent_t *e = timed ? &h->ev[(size_t)h->ev_used * 3] : nullptr;
    if (int rc = launch_inputs(h, st)) return rc;
    if (timed) HIPCHK(hipEventRecord(e[0], st));
    if (int rc = tt ? launch_move_tt<false>(h, t, st) : launch_move(h, t, st)) return rc;
    if (timed) HIPCHK(hipEventRecord(e[1], st));
    if (int rc = launch_advance(h, t, st)) return rc;
    if (timed) {
      HIPCHK(hipEventRecord(e[2], st));
      h->ev_weight[h->ev_used] = 1;
      ++h->ev_used;
    }
  }
  return TFX_OK;
}

}  // namespace

int step_chunk(tfx_handle h, int n_ticks, hipStream_t st);

extern "C" int tfx_agent_step(tfx_handle h, int32_t n_ticks, int32_t remi, float *aobs, float *areward,
                              uint8_t *adone, void *stream) {
  if (int rc = check_handle(h, true)) return rc;
  if (n_ticks < 1) return fail(TFX_EINVAL, "n_ticks < 1");
  if (h->action_per_tick)
    return fail(TFX_EINVAL, "the fused agent step holds ONE action for all its ticks (bind the action buffer "
                            "with per_tick = 0)");
  hipStream_t st = (hipStream_t)stream;
  const Dev &d = h->d;
  // a batch whose halves still fill the chip: two halves on two streams (k_tail of one under the pass of the other)
  const bool split = !res_usable(h, n_ticks) && !h->poisson && split_usable(h, n_ticks);
  if (split) {
    if (int rc = ensure_split(h)) return rc;
  }
  if (!h->use_graph || split) {
    long long nf = 0, np = 0;
    const int rc = agent_sequence(h, n_ticks, remi, aobs, areward, adone, st, nf, np, split);
    if (rc == TFX_OK) {
      h->fused_ticks += nf;
      h->pair_ticks += np;
      if (split) h->split_ticks += n_ticks;
    }
    return rc;
  }
  // one graph per distinct launch sequence: everything baked into kernel arguments is in the key
  char key[640];
  snprintf(key, sizeof key, "%llu|%u|%u|%d|%d%d%d|%ld|%d|%d|%p|%p|%p|%p|%d|%d|%p|%d|%d|%p|%p|%p|%p|%p|%p|%p|%p",
           h->input_gen, h->ps.seed_lo, h->ps.seed_hi, h->ps.n_cdf, (int)h->poisson,
           (int)h->greedy, h->greedy_spacing, d.spawn_stride, n_ticks, remi, (void *)aobs,
           (void *)areward, (void *)adone, (const void *)d.action, d.action_mode, d.action_period,
           (const void *)d.spawn, d.spawn_mode, d.spawn_period, (void *)d.xv, (void *)d.w, (void *)d.obs,
           (void *)d.rewards, (void *)d.leading, (void *)d.lastcar, (void *)d.waiting, (void *)d.done_tick);
  if (!h->ag_exec || h->ag_key != key) {
    if (h->ag_exec) { (void)hipGraphExecDestroy(h->ag_exec); h->ag_exec = nullptr; }
    if (h->ag_graph) { (void)hipGraphDestroy(h->ag_graph); h->ag_graph = nullptr; }
    if (!h->ag_stream) HIPCHK(hipStreamCreateWithFlags(&h->ag_stream, hipStreamNonBlocking));
    if (h->grid_move == 0 && !res_usable(h, n_ticks)) {  // size the move grid outside the capture (occupancy queries)
      if (int rc = launch_move_probe(h)) return rc;
    }
    if (!res_usable(h, n_ticks) && h->grid_adv == 0) {
      h->size_only = true;
      (void)launch_advance(h, 0, nullptr);
      h->size_only = false;
    }
    if (!res_usable(h, n_ticks) && pairs_usable(h)) {
      h->size_only = true;
      (void)launch_move_tt<true, true>(h, 0, nullptr);
      (void)launch_move_tt<false, true>(h, 0, nullptr);
      (void)launch_tail(h, 0, nullptr, true);
      h->size_only = false;
      (void)edge_grid(h);
    }
    HIPCHK(hipStreamBeginCapture(h->ag_stream, hipStreamCaptureModeThreadLocal));
    const int rc = agent_sequence(h, n_ticks, remi, aobs, areward, adone, h->ag_stream, h->ag_fused, h->ag_pair);
    hipGraph_t g = nullptr;
    const hipError_t ce = hipStreamEndCapture(h->ag_stream, &g);
    if (rc != TFX_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (ce != hipSuccess) return fail(TFX_EDEVICE, "hipStreamEndCapture: %s", hipGetErrorString(ce));
    h->ag_graph = g;
    HIPCHK(hipGraphInstantiate(&h->ag_exec, g, nullptr, nullptr, 0));
    h->ag_key = key;
  }
  HIPCHK(hipGraphLaunch(h->ag_exec, st));
  h->fused_ticks += h->ag_fused;  // (the capture ran no kernel: every replay counts)
  h->pair_ticks += h->ag_pair;
  return TFX_OK;
}

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
  if (cfg->layout != 0 && cfg->layout != 1) return fail(TFX_EINVAL, "layout must be 0 (ring) or 1 (transposed)");
  const int n_arch = cfg->n_archetypes <= 1 ? 1 : cfg->n_archetypes;
  if (n_arch > TFX_MAX_ARCH) return fail(TFX_EINVAL, "at most %d archetype rows", TFX_MAX_ARCH);
  bool het = n_arch > 1;
  float arch_rows[TFX_MAX_ARCH][8] = {};
  for (int a = 0; a < n_arch; ++a) {
    const float single[8] = {cfg->car_v, cfg->car_l, cfg->car_a, cfg->car_delta, cfg->car_v0, cfg->car_b, cfg->car_T, cfg->car_s0};
    memcpy(arch_rows[a], cfg->n_archetypes >= 1 ? cfg->arch[a] : single, sizeof single);
    const float delta = arch_rows[a][3];
    if (!(delta >= 1.0f && delta <= 8.0f) || delta != (float)(int)delta)
      return fail(TFX_EINVAL, "archetype %d: delta = %g - only integers 1..8 have a bit-exact power (oracle powi_cr)", a, delta);
    if (delta != 4.0f) het = true;
    if (!(arch_rows[a][2] > 0.0f) || !(arch_rows[a][5] > 0.0f) || !(arch_rows[a][4] > 0.0f))
      return fail(TFX_EINVAL, "archetype %d: a, b and v0 must be > 0", a);
  }
  if (het && (cfg->layout != 1 || cfg->planes != 3))
    return fail(TFX_EINVAL, "heterogeneous cars (several archetypes, or delta != 4) need layout = 1 and planes = 3");
  if (!(cfg->length > 0.0f) || !(cfg->rate > 0.0f)) return fail(TFX_EINVAL, "length and rate must be > 0");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev < 1) return fail(TFX_EDEVICE, "no HIP device");
  tfx_handle_s *h = new (std::nothrow) tfx_handle_s();
  if (!h) return fail(TFX_ENOMEM, "out of host memory");
  h->cfg = *cfg;
  h->het = het;
  if (!het && cfg->n_archetypes == 1) {  // one ordinary row given through the table: it IS the archetype
    h->cfg.car_v = arch_rows[0][0]; h->cfg.car_l = arch_rows[0][1]; h->cfg.car_a = arch_rows[0][2];
    h->cfg.car_delta = arch_rows[0][3]; h->cfg.car_v0 = arch_rows[0][4]; h->cfg.car_b = arch_rows[0][5];
    h->cfg.car_T = arch_rows[0][6]; h->cfg.car_s0 = arch_rows[0][7];
    cfg = &h->cfg;
  }
  build_tables(h);
  build_slots(h);
  if (const char *mv = getenv("TFX_MOVE_VARIANT")) h->move_variant = atoi(mv);
  if (const char *gr = getenv("TFX_GRAPH")) h->use_graph = atoi(gr) != 0;
  if (const char *pv = getenv("TFX_PAIRS")) h->pairs = atoi(pv);
  if (const char *tv = getenv("TFX_TAIL")) h->tail = atoi(tv);
  if (const char *sv = getenv("TFX_SPLIT")) h->split = atoi(sv);
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
  d.n_entry = (int)h->h_entry.size();
  d.obs_len = 2 * d.r + 2 * d.I;
  d.yellow = cfg->yellow_ticks;
  d.learn_switch = cfg->learn_switch;
  d.validate = cfg->validate;
  d.env_off = cfg->env_id_offset;
  d.layout = cfg->layout;
  d.length = cfg->length;
  d.rate = cfg->rate;
  d.car_v = cfg->car_v; d.car_l = cfg->car_l; d.car_a = cfg->car_a; d.car_v0 = cfg->car_v0;
  d.car_b = cfg->car_b; d.car_T = cfg->car_T; d.car_s0 = cfg->car_s0;
  d.two_sab = 2.0f * sqrtf(cfg->car_a * cfg->car_b);  // 2 * np.sqrt(a*b) (traffic_env.py:54)
  d.eps = cfg->eps;
  d.r_two_sab = 1.0f / d.two_sab;
  d.r_v0 = 1.0f / cfg->car_v0;
  d.thresh = cfg->thresh;
  d.near_end = cfg->length - cfg->detect_dist;
  d.ovf_pen = cfg->overflow_penalty;
  if ((long)d.E * d.R > 0x7fffffffL / 4) { delete h; return fail(TFX_EINVAL, "E*R too large"); }

  const int cars = d.C - 2;
  h->wpr = cars <= 64 ? 1 : (cars <= 128 ? 2 : 4);
  h->grid_move = 0;  // sized at the first launch from the kernel's occupancy

  // tables
  const size_t R = (size_t)d.R;
  const size_t n_slots = h->h_slot_road.size();
  if (hipMalloc((void **)&h->dev_tables, (4 * R + n_slots) * sizeof(int)) != hipSuccess) {
    delete h;
    return fail(TFX_ENOMEM, "hipMalloc(tables) failed");
  }
  if (hipMemcpy(h->dev_tables, h->h_nexts.data(), R * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(h->dev_tables + R, h->h_pred.data(), R * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(h->dev_tables + 2 * R, h->h_entry_idx.data(), R * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(h->dev_tables + 3 * R, h->h_road_slot.data(), R * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(h->dev_tables + 4 * R, h->h_slot_road.data(), n_slots * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(h->dev_tables);
    delete h;
    return fail(TFX_EDEVICE, "uploading the road tables failed");
  }
  d.nexts = h->dev_tables;
  d.pred = h->dev_tables + R;
  d.entry_idx = h->dev_tables + 2 * R;
  d.road_slot = h->dev_tables + 3 * R;
  d.slot_road = h->dev_tables + 4 * R;
  d.G = h->tiles_per_env;

  // scratch
  const size_t ER = (size_t)d.E * R;
  size_t off = 0;
  const size_t o_rec = off;   off = align_up(off + ER * sizeof(int4), 256);
  const size_t o_rec2 = off;  off = align_up(off + (d.layout == 1 ? ER * sizeof(int4) : 0), 256);
  const size_t o_tail = off;  off = align_up(off + ER * sizeof(float), 256);
  const size_t o_flag = off;  off = align_up(off + (size_t)d.E * sizeof(int), 256);
  const size_t o_risk = off;  off = align_up(off + (size_t)d.E * sizeof(int), 256);
  d.trows = d.C - 2;  // (padding the tile stride off the power of two was measured: slightly slower)
  const size_t n_tpairs = (size_t)d.E * d.G * (size_t)d.trows * 64;  // (x, v) pairs of a transposed array
  // outbox: TFX_KP rows per tile (the cars a road hands over in a tick; round 1 kept a T-sized one)
  const size_t n_opairs = (size_t)d.E * d.G * KP * 64;
  const size_t o_outb = off;  off = align_up(off + (d.layout == 1 ? n_opairs * sizeof(float2) : 0), 256);
  const size_t o_outw = off;  off = align_up(off + (d.layout == 1 && cfg->planes == 3 ? n_opairs * sizeof(float) : 0), 256);
  const size_t o_lead = off;  off = align_up(off + ER * sizeof(float), 256);
  const size_t o_taila = off; off = align_up(off + (het ? ER * sizeof(int) : 0), 256);
  const size_t o_misc = off;  off = align_up(off + 128, 256);
  const size_t o_veh = off;   off = align_up(off + (size_t)VEH_SLOTS * VEH_STRIDE * sizeof(unsigned long long), 256);
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
  d.rec2 = (int4 *)(base + o_rec2);
  d.tailx = (float *)(base + o_tail);
  d.env_flag = (int *)(base + o_flag);
  d.env_risk = (int *)(base + o_risk);
  d.outb = (float2 *)(base + o_outb);
  d.outw = (float *)(base + o_outw);
  d.leadx = (float *)(base + o_lead);
  d.het = het ? 1 : 0;
  d.taila = (int *)(base + o_taila);
  if (het) {
    float tab[TFX_MAX_ARCH][ARCH_W] = {};
    for (int a = 0; a < n_arch; ++a) {
      const float *r = arch_rows[a];  // v, l, a, delta, v0, b, T, s0
      tab[a][AR_L] = r[1]; tab[a][AR_A] = r[2]; tab[a][AR_V0] = r[4]; tab[a][AR_T] = r[6]; tab[a][AR_S0] = r[7];
      tab[a][AR_2SAB] = 2.0f * sqrtf(r[2] * r[5]);  // 2 * np.sqrt(a*b) (traffic_env.py:54)
      tab[a][AR_DELTA] = r[3]; tab[a][AR_V] = r[0];
    }
    if (hipMalloc((void **)&h->dev_arch, sizeof tab) != hipSuccess ||
        hipMemcpy(h->dev_arch, tab, sizeof tab, hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipFree(h->dev_tables);
      (void)hipFree(h->dev_scratch);
      delete h;
      return fail(TFX_ENOMEM, "uploading the archetype table failed");
    }
    d.arch_tab = h->dev_arch;
  }
  h->n_tpairs = n_tpairs;
  d.veh = (unsigned long long *)(base + o_veh);
  d.tickA = (int *)(base + o_misc + 16);
  d.tickB = (int *)(base + o_misc + 32);
  d.agent_first = (const int *)(base + o_misc + 48);
  d.risk_any = (int *)(base + o_misc + 56);
  h->tick2 = (int *)(base + o_misc + 64);
  // reciprocal division is used only if it is exact for this handle's constants on the whole
  // admitted numerator domain (2 x ~2^31 quotients, a few milliseconds; TFX_FASTDIV=0 disables)
  d.fastdiv = 0;
  {
    const char *fd = getenv("TFX_FASTDIV");
    if (!fd || atoi(fd) != 0) {
      unsigned long long *bad = (unsigned long long *)(base + o_misc);  // scratch word, zero at this point
      hipLaunchKernelGGL(k_div_selftest, dim3(h->n_cu * 8), dim3(256), 0, 0, d.two_sab, d.r_two_sab,
                         TFX_FASTDIV_A_LO, TFX_FASTDIV_A_HI, bad);
      hipLaunchKernelGGL(k_div_selftest, dim3(h->n_cu * 8), dim3(256), 0, 0, cfg->car_v0, d.r_v0,
                         TFX_FASTDIV_V_LO, TFX_FASTDIV_V_HI, bad);
      unsigned long long nbad = 1;
      if (hipMemcpy(&nbad, bad, sizeof nbad, hipMemcpyDeviceToHost) == hipSuccess && nbad == 0) d.fastdiv = 1;
      (void)hipMemset(bad, 0, sizeof nbad);
      h->div_mismatches = nbad;
    }
  }
  {
    unsigned *bad = (unsigned *)(base + o_misc);  // (cleared again above)
    hipLaunchKernelGGL(k_max_selftest, dim3(1), dim3(1), 0, 0, bad, 0.0f, -0.0f);
    unsigned nbad = 1;
    if (hipMemcpy(&nbad, bad, sizeof nbad, hipMemcpyDeviceToHost) == hipSuccess && nbad == 0) d.fastmax = 1;
    (void)hipMemset(bad, 0, 8);
  }
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
  if (h->ag_exec) (void)hipGraphExecDestroy(h->ag_exec);
  if (h->ag_graph) (void)hipGraphDestroy(h->ag_graph);
  if (h->ag_stream) (void)hipStreamDestroy(h->ag_stream);
  if (h->split_stream) {
    (void)hipStreamSynchronize(h->split_stream);
    (void)hipStreamDestroy(h->split_stream);
    (void)hipEventDestroy(h->split_fork);
    (void)hipEventDestroy(h->split_join);
    (void)hipEventDestroy(h->split_stagger);
  }
  if (h->dev_ps) (void)hipFree(h->dev_ps);
  if (h->dev_greedy) (void)hipFree(h->dev_greedy);
  if (h->dev_tables) (void)hipFree(h->dev_tables);
  if (h->dev_scratch) (void)hipFree(h->dev_scratch);
  if (h->dev_arch) (void)hipFree(h->dev_arch);
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
  if (!b->xv || !b->leading || !b->lastcar || !b->obs || !b->rewards || !b->waiting ||
      !b->passed_dst || !b->done_tick)
    return fail(TFX_EINVAL, "xv, leading, lastcar, obs, rewards, waiting, passed_dst and done_tick are required");
  if (h->cfg.planes == 3 && !b->w) return fail(TFX_EINVAL, "planes = 3 needs the w buffer");
  if (((uintptr_t)b->xv & 15u) != 0) return fail(TFX_EINVAL, "xv must be 16-byte aligned");
  if (h->cfg.validate && (!b->n_trips || (b->trip_times && b->trip_cap < 1)))
    return fail(TFX_EINVAL, "validate mode needs n_trips (and trip_cap >= 1 with trip_times)");
  Dev &d = h->d;
  d.xv = reinterpret_cast<float2 *>(b->xv); d.w = b->w;
  d.leading = b->leading; d.lastcar = b->lastcar; d.obs = b->obs;
  d.rewards = b->rewards; d.waiting = b->waiting; d.passed_dst = b->passed_dst;
  d.done_tick = b->done_tick; d.trip_times = b->trip_times; d.n_trips = b->n_trips;
  d.trip_cap = b->trip_cap;
  h->bound = true;
  ++h->input_gen;
  return d.w ? res_configure<true>(h) : res_configure<false>(h);
}

int tfx_reset(tfx_handle h, const int32_t *phase_init, void *stream) {
  if (int rc = check_handle(h, true)) return rc;
  if (!phase_init) return fail(TFX_EINVAL, "phase_init is required");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipMemsetAsync(h->d.tickA, 0, sizeof(int), st));
  HIPCHK(hipMemsetAsync(h->d.tickB, 0, sizeof(int), st));
  hipLaunchKernelGGL(k_reset, dim3(grid_for((long)h->d.E * h->d.R, h->n_cu)), dim3(256), 0, st, h->d, phase_init,
                     (const uint8_t *)nullptr);
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

int tfx_reset_envs(tfx_handle h, const int32_t *phase_init, const uint8_t *mask, void *stream) {
  if (int rc = check_handle(h, true)) return rc;
  if (!phase_init || !mask) return fail(TFX_EINVAL, "phase_init and mask are required");
  hipLaunchKernelGGL(k_reset, dim3(grid_for((long)h->d.E * h->d.R, h->n_cu)), dim3(256), 0, (hipStream_t)stream,
                     h->d, phase_init, mask);
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
  ++h->input_gen;
  h->greedy = false;
  d.greedy_spacing = 0;
  d.greedy_act = nullptr;
  if (mode == TFX_ACTION_GREEDY) {
    if (period < 1) return fail(TFX_EINVAL, "greedy spacing must be >= 1");
    if (!h->dev_greedy) {
      HIPCHK(hipMalloc((void **)&h->dev_greedy, (size_t)d.E * d.I * sizeof(int)));
      HIPCHK(hipMemset(h->dev_greedy, 0, (size_t)d.E * d.I * sizeof(int)));
    }
    h->greedy = true;
    h->greedy_spacing = period;
    d.greedy_spacing = period;
    d.greedy_act = h->dev_greedy;
    d.action = h->dev_greedy;
    d.action_stride = 0;
    d.action_mode = TFX_ACTION_BUFFER;
    h->action_per_tick = 0;
    return TFX_OK;
  }
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
  ++h->input_gen;
  h->poisson = false;
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

int tfx_set_spawn_archetypes(tfx_handle h, const uint8_t *dev, int32_t per_road, int32_t per_tick) {
  if (int rc = check_handle(h, false)) return rc;
  if (!h->het) return fail(TFX_ESTATE, "the handle has a single archetype");
  if (dev && per_road < 1) return fail(TFX_EINVAL, "per_road must be >= 1");
  Dev &d = h->d;
  ++h->input_gen;
  d.spawn_arch = dev;
  d.spawn_arch_S = dev ? per_road : 0;
  d.spawn_arch_stride = (dev && per_tick) ? (long)d.E * d.n_entry * per_road : 0;
  return TFX_OK;
}

int tfx_set_poisson(tfx_handle h, double cars_per_tick, uint64_t seed, const uint32_t *cdf, int32_t n_cdf) {
  if (int rc = check_handle(h, false)) return rc;
  if (!(cars_per_tick > 0.0)) return fail(TFX_EINVAL, "cars_per_tick must be > 0");
  if (!cdf || n_cdf < 1 || n_cdf > 65536) return fail(TFX_EINVAL, "gap table missing or too long");
  Dev &d = h->d;
  if (d.n_entry < 1) return fail(TFX_EINVAL, "no entry roads");
  if (h->het) return fail(TFX_EINVAL, "the on-device Poisson stream draws no archetype rows: feed heterogeneous cars through "
                                      "tfx_set_spawns + tfx_set_spawn_archetypes");
  ++h->input_gen;
  if (h->dev_ps) { (void)hipFree(h->dev_ps); h->dev_ps = nullptr; }
  // rows of E x n_entry counts: tfx_step generates that many ticks per launch (at most 64, at most ~32 MB)
  long rows = ((long)32 << 20) / ((long)d.E * d.n_entry * 4);
  h->poisson_rows = (int)(rows < 1 ? 1 : (rows > 64 ? 64 : rows));
  const size_t n_counts = (size_t)h->poisson_rows * d.E * d.n_entry;
  const size_t bytes = (n_counts + 2 * (size_t)d.E + (size_t)n_cdf) * 4;
  HIPCHK(hipMalloc(&h->dev_ps, bytes));
  HIPCHK(hipMemset(h->dev_ps, 0, bytes));
  int *base = (int *)h->dev_ps;
  h->ps.counts = base;
  h->ps.gap_left = base + n_counts;
  h->ps.draws = (unsigned *)(base + n_counts + d.E);
  h->ps.cdf = (const unsigned *)(base + n_counts + 2 * (size_t)d.E);
  h->ps.n_cdf = n_cdf;
  h->ps.seed_lo = (unsigned)seed;
  h->ps.seed_hi = (unsigned)(seed >> 32);
  HIPCHK(hipMemset(h->ps.gap_left, 0xff, (size_t)d.E * 4));  // -1: first gap not drawn yet
  HIPCHK(hipMemcpy((void *)h->ps.cdf, cdf, (size_t)n_cdf * 4, hipMemcpyHostToDevice));
  d.spawn = h->ps.counts;
  d.spawn_stride = 0;
  d.spawn_mode = TFX_SPAWN_COUNTS;
  h->spawn_per_tick = 0;
  h->poisson = true;
  return TFX_OK;
}

int tfx_step(tfx_handle h, int32_t n_ticks, void *stream) {
  if (int rc = check_handle(h, true)) return rc;
  if (n_ticks < 0) return fail(TFX_EINVAL, "n_ticks < 0");
  hipStream_t st = (hipStream_t)stream;
  // envs that fit a compute unit's LDS: all the ticks of the call in one launch (tfx_resident.hpp)
  if (n_ticks > 0 && res_usable(h, n_ticks)) {
    const bool timed = h->prof && h->ev_used < h->ev_ticks;
    hipEvent_t *e = timed ? &h->ev[(size_t)h->ev_used * 3] : nullptr;
    if (timed) HIPCHK(hipEventRecord(e[0], st));
    if (int rc = launch_res(h, n_ticks, st)) return rc;
    h->fused_ticks += n_ticks;
    if (timed) {
      HIPCHK(hipEventRecord(e[1], st));
      HIPCHK(hipEventRecord(e[2], st));
      h->ev_weight[h->ev_used] = n_ticks;
      ++h->ev_used;
    }
    return TFX_OK;
  }
  if (int rc = launch_greedy(h, st)) return rc;  // (the decision of the call's first tick; later ones: advance_item)
  if (h->poisson && n_ticks > 0) {
    // the arrivals of the whole call (in chunks of the rows the count buffer holds) in ONE launch each: they depend
    // on nothing but the stream, and a launch per tick was the longest one of a cfg4 tick
    int rc = TFX_OK;
    for (int done = 0; done < n_ticks && rc == TFX_OK;) {
      const int chunk = n_ticks - done < h->poisson_rows ? n_ticks - done : h->poisson_rows;
      rc = launch_poisson(h, chunk, st);
      h->d.spawn_stride = (long)h->d.E * h->d.n_entry;
      if (rc == TFX_OK) rc = step_chunk(h, chunk, st);
      h->d.spawn_stride = 0;
      done += chunk;
    }
    return rc;
  }
  return step_chunk(h, n_ticks, st);
}

}  // extern "C"

// n_ticks ticks on the per-tick kernels, the env range in two halves on two streams where that pays
int step_chunk(tfx_handle h, int n_ticks, hipStream_t st) {
  if (split_usable(h, n_ticks)) {
    // fork: the handle's own stream takes the second half of the envs, the caller's stream the first
    if (int rc = ensure_split(h)) return rc;
    if (h->grid_tt[1] == 0 || h->grid_tt[0] == 0 || h->grid_tail == 0) {  // grids are sized for the whole range
      h->size_only = true;
      (void)launch_move_tt<true>(h, 0, nullptr);
      (void)launch_move_tt<false>(h, 0, nullptr);
      (void)launch_tail(h, 0, nullptr);
      (void)launch_advance(h, 0, nullptr);
      h->size_only = false;
    }
    const Dev whole = h->d;
    const int n0 = whole.E / 2;
    const long long pair0 = h->pair_ticks, tail0 = h->tail_ticks;
    HIPCHK(hipEventRecord(h->split_fork, st));
    HIPCHK(hipStreamWaitEvent(h->split_stream, h->split_fork, 0));
    hipLaunchKernelGGL(k_clock_copy, dim3(1), dim3(1), 0, h->split_stream, whole.tickA, whole.tickB, h->tick2);
    int rc = hipGetLastError() == hipSuccess ? TFX_OK : fail(TFX_EDEVICE, "k_clock_copy launch failed");
    for (int half = 0; half < 2 && rc == TFX_OK; ++half) {
      h->d = half == 0 ? sub_dev(h, 0, n0, nullptr) : sub_dev(h, n0, whole.E - n0, h->tick2);
      h->split_half = half;
      h->split_first = true;
      rc = step_range(h, n_ticks, half == 0 ? st : h->split_stream);
      h->d = whole;
    }
    h->split_half = -1;
    if (rc != TFX_OK) return rc;
    h->pair_ticks = pair0 + (h->pair_ticks - pair0) / 2;  // (both halves counted them)
    h->tail_ticks = tail0 + (h->tail_ticks - tail0) / 2;
    h->split_ticks += n_ticks;
    HIPCHK(hipEventRecord(h->split_join, h->split_stream));
    HIPCHK(hipStreamWaitEvent(st, h->split_join, 0));
    return TFX_OK;
  }
  return step_range(h, n_ticks, st);
}

extern "C" {

int tfx_move_cars(tfx_handle h, void *stream) {
  if (int rc = check_handle(h, true)) return rc;
  if (int rc = launch_greedy(h, (hipStream_t)stream)) return rc;
  if (int rc = launch_inputs(h, (hipStream_t)stream)) return rc;
  return pairs_usable(h) ? launch_move_tt<false>(h, 0, (hipStream_t)stream) : launch_move(h, 0, (hipStream_t)stream);
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
  std::vector<unsigned long long> slots((size_t)VEH_SLOTS * VEH_STRIDE);
  HIPCHK(hipMemcpy(slots.data(), h->d.veh, slots.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  unsigned long long v = 0;
  for (int i = 0; i < VEH_SLOTS; ++i) v += slots[(size_t)i * VEH_STRIDE];
  *out = (uint64_t)v;
  return TFX_OK;
}

int tfx_reset_counters(tfx_handle h, void *stream) {
  if (int rc = check_handle(h, false)) return rc;
  HIPCHK(hipMemsetAsync(h->d.veh, 0, (size_t)VEH_SLOTS * VEH_STRIDE * sizeof(unsigned long long), (hipStream_t)stream));
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
  h->ev_weight.assign((size_t)max_ticks, 1);
  for (hipEvent_t &e : h->ev) HIPCHK(hipEventCreate(&e));
  h->ev_ticks = max_ticks;
  return TFX_OK;
}

int tfx_profile_read(tfx_handle h, double *move_ms, double *advance_ms, int32_t *n_ticks) {
  if (int rc = check_handle(h, false)) return rc;
  double mv = 0.0, ad = 0.0;
  int ticks = 0;
  for (int i = 0; i < h->ev_used; ++i) {
    float a = 0.f, b = 0.f;
    HIPCHK(hipEventSynchronize(h->ev[(size_t)i * 3 + 2]));
    HIPCHK(hipEventElapsedTime(&a, h->ev[(size_t)i * 3], h->ev[(size_t)i * 3 + 1]));
    HIPCHK(hipEventElapsedTime(&b, h->ev[(size_t)i * 3 + 1], h->ev[(size_t)i * 3 + 2]));
    mv += a;
    ad += b;
    ticks += h->ev_weight[i];
  }
  if (move_ms) *move_ms = mv;
  if (advance_ms) *advance_ms = ad;
  if (n_ticks) *n_ticks = ticks;
  h->ev_used = 0;
  return TFX_OK;
}

int tfx_xv_pairs(tfx_handle h, int64_t *pairs) {
  if (int rc = check_handle(h, false)) return rc;
  if (!pairs) return fail(TFX_EINVAL, "pairs is null");
  *pairs = h->d.layout == 1 ? (int64_t)h->n_tpairs : (int64_t)h->d.E * h->d.R * h->d.C;
  return TFX_OK;
}

int tfx_export_ring(tfx_handle h, float *ring_xv, float *ring_w, uint8_t *ring_a, void *stream) {
  if (int rc = check_handle(h, true)) return rc;
  if (h->d.layout != 1) return fail(TFX_ESTATE, "the handle already uses the ring layout");
  if (!ring_xv) return fail(TFX_EINVAL, "ring_xv is null");
  hipLaunchKernelGGL(k_export_ring, dim3(grid_for((long)h->d.E * h->d.R, h->n_cu)), dim3(256), 0,
                     (hipStream_t)stream, h->d, reinterpret_cast<float2 *>(ring_xv), ring_w, ring_a);
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

int tfx_import_ring(tfx_handle h, const float *ring_xv, const float *ring_w, const uint8_t *ring_a, void *stream) {
  if (int rc = check_handle(h, true)) return rc;
  if (h->d.layout != 1) return fail(TFX_ESTATE, "the handle already uses the ring layout");
  if (!ring_xv) return fail(TFX_EINVAL, "ring_xv is null");
  hipLaunchKernelGGL(k_import_ring, dim3(grid_for((long)h->d.E * h->d.R, h->n_cu)), dim3(256), 0,
                     (hipStream_t)stream, h->d, reinterpret_cast<const float2 *>(ring_xv), ring_w, ring_a);
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

int tfx_fastdiv_status(tfx_handle h, int32_t *enabled, uint64_t *mismatches) {
  if (int rc = check_handle(h, false)) return rc;
  if (enabled) *enabled = h->d.fastdiv;
  if (mismatches) *mismatches = h->div_mismatches;
  return TFX_OK;
}

int tfx_fused_ticks(tfx_handle h, int64_t *ticks, int32_t *capable) {
  if (int rc = check_handle(h, false)) return rc;
  if (ticks) *ticks = h->fused_ticks;
  if (capable) *capable = h->res_epb > 0 ? 1 : 0;
  return TFX_OK;
}

int tfx_pair_ticks(tfx_handle h, int64_t *ticks) {
  if (int rc = check_handle(h, false)) return rc;
  if (ticks) *ticks = h->pair_ticks;
  return TFX_OK;
}

int tfx_tail_ticks(tfx_handle h, int64_t *ticks) {
  if (int rc = check_handle(h, false)) return rc;
  if (ticks) *ticks = h->tail_ticks;
  return TFX_OK;
}

int tfx_split_ticks(tfx_handle h, int64_t *ticks) {
  if (int rc = check_handle(h, false)) return rc;
  if (ticks) *ticks = h->split_ticks;
  return TFX_OK;
}

const char *tfx_step_kernel(tfx_handle h) { return h ? h->step_kernel : ""; }

int tfx_launch_info(tfx_handle h, int32_t *grid, int32_t *block, int32_t *waves_per_road) {
  if (int rc = check_handle(h, false)) return rc;
  if (h->grid_move == 0) return fail(TFX_ESTATE, "no move kernel has been launched yet");
  if (grid) *grid = h->grid_move;
  if (block) *block = 256;
  if (waves_per_road) *waves_per_road = h->wpr;
  return TFX_OK;
}

}  // extern "C"

// host-only: replay of the reference's seeded arrival generators for many envs
#include "tfx_arrivals.cpp"
