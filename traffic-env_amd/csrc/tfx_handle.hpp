// tfx_handle.hpp - the handle behind the C ABI (include/tfx.h): configuration, device parameter block, launch
// geometry, optional per-kernel timing, the tables of GridRoad (roadgraph.py:26-64) and the storage-slot order.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "tfx.h"
#include "tfx_common.hpp"
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
  int tt_segs = 0;            // TFX_TT_SEGS: 2 / 4 / 8 forces k_move_tts's wavefronts per tile (0: by launch size)
  int tt_seg = 1;             // TFX_TT_SEG: 0 never, 1 launches that leave wave slots empty, 2 whenever the form exists
  int grid_adv = 0;
  int grid_tail = 0, grid_tail_half = 0;  // k_tail: workgroups of 256 lanes / of 128 (the halves of a split call)
  int tail_threads = 256, tail_threads_half = 128;
  int tail = 1;               // k_tail after a two-tick pass (tfx_tail.hpp): TFX_TAIL=0 never, 2 at any batch size
  // calls of pairs run as TWO halves of the env range, the second on a stream of the handle's own: the latency-bound
  // per-road launch of one half (k_tail) runs under the other half's car pass (split_usable)
  int split = 1;              // TFX_SPLIT=0 never, 2 at any batch size
  hipStream_t split_stream = nullptr;
  int split_prio = 0;  // priority level split_stream was created with
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
  int res_lpr = 1;            // lanes per road: 1, 2, 4, or 3 = two and four on the roads without a predecessor
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
  // tfx_step calls of launch-bound handles replay a captured graph as well (step_graph, tfx_hip.hip)
  hipGraph_t st_graph = nullptr;
  hipGraphExec_t st_exec = nullptr;
  std::string st_key;
  long long st_pair = 0, st_tail = 0;  // what one replay adds to pair_ticks / tail_ticks
  // bumped by every call that changes something a captured kernel argument was built from (bound
  // buffers, action / spawn rules, the Poisson stream): part of the graph key, so a stale graph is
  // never replayed even when a re-allocated buffer lands on the address the old one had
  unsigned long long input_gen = 0;
  bool use_graph = true;  // TFX_GRAPH=0 disables
  bool size_only = false;
  int fail_after = 0;  // tfx_debug_fail_after: the n-th launch from now fails (error-path tests); 0 = off
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

// fault injection for the error-path tests (tfx_debug_fail_after): true when THIS launch is the one to fail
bool inject_failure(tfx_handle h) {
  if (h->size_only || h->fail_after <= 0) return false;
  return --h->fail_after == 0;
}
#define TFX_INJECT(h)                                                                              \
  do {                                                                                             \
    if (inject_failure(h)) return fail(TFX_EDEVICE, "injected launch failure (tfx_debug_fail_after)"); \
  } while (0)

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int check_handle(tfx_handle h, bool need_bound) {
  if (!h) return fail(TFX_EINVAL, "null handle");
  if (need_bound && !h->bound) return fail(TFX_ESTATE, "tfx_bind_buffers has not been called");
  return TFX_OK;
}

}  // namespace
