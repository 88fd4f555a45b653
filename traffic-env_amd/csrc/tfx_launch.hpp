// tfx_launch.hpp - which kernel moves the cars of a handle, with what grid, and the launch helpers of the per-tick
// kernels (host side; included by tfx_hip.hip).
#pragma once
#include "tfx_handle.hpp"
#include "tfx_move_generic.hpp"
#include "tfx_move_dma.hpp"
#include "tfx_move_t.hpp"
#include "tfx_move_ts.hpp"
#include "tfx_move_tt.hpp"
#include "tfx_resident.hpp"
#include "tfx_advance.hpp"
#include "tfx_tail.hpp"
#include "tfx_move_tts.hpp"

namespace {

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
  TFX_INJECT(h);
  hipLaunchKernelGGL(kern, dim3(h->grid_move), dim3(256), h->move_lds, st, h->d, tidx);
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

template <int WPR>
int launch_generic(tfx_handle h, int tidx, hipStream_t st) {
  h->step_kernel = "k_move";
  if (h->grid_move == 0) h->grid_move = move_grid(h, k_move<WPR>, 256 / (64 * WPR));
  if (h->size_only) return TFX_OK;
  TFX_INJECT(h);
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
    TFX_INJECT(h);
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
    TFX_INJECT(h);
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
    // long roads on a launch of at most ~two tiles per CU: sixteen segments of 8 cars instead of four of 32
    // (cfg4 x 1 env closed loop, eight of 16: 38.4 -> 28.6 us per tick)
    // (sixteen of 8: 29.6 -> 28.8 us per tick at one env, 39.2 -> 37.7 at two - what is left is the launches' own latency)
    if (cap <= 128 && cap > 64 && tiles <= (long)h->n_cu * 2) { h->step_kernel = "k_move_ts"; return gs(k_move_ts<8, false, 16>, 1024); }
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
  // (LPR = 3: two lanes per road and two more on each of the 2 (m + n) roads without a predecessor; 32 spare columns
  // for the lanes past the last env)
  const int n_ent = 2 * (h->cfg.m + h->cfg.n);
  const int threads = ((LPR == 3 ? epb * (2 * d.R + 2 * n_ent) : LPR * epb * d.R) + 63) / 64 * 64;
  if (threads > RES_MAX_THREADS) return false;
  const int cols = LPR == 3 ? epb * d.R + 32 : threads / LPR;
  size_t lds = res_lds_bytes(cols, d.C, epb, d.I, d.n_entry, W);
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
  // Lanes per road.  Two by default.  FOUR (8-car chains instead of 16 at cfg1) while the launch leaves the chip half
  // empty - there a tick is as long as its longest chain: cfg1 x 256 envs, a 10-tick call 84 -> 64 us, a fused decision
  // 74 -> 59; at 1024 envs the five wavefronts an env then takes no longer fit beside each other (16 per CU at this
  // kernel's 128 registers) and the call takes 195 us instead of 106.  TFX_RES_LPR = 1 / 2 / 4 forces.
  // Otherwise the MIXED form (3): two lanes per road, four on the roads cars enter the map on - the longest chains of a
  // loaded network - in the same number of wavefronts per env as two everywhere (cfg1: 160 + 32 lanes = 3 wavefronts).
  const long waves4 = ((long)4 * d.R + 63) / 64;
  int lpr_max = ((long)d.E * waves4 <= (long)h->n_cu * 8) ? 4 : 3;
  if (const char *lv = getenv("TFX_RES_LPR")) lpr_max = (atoi(lv) >= 1 && atoi(lv) <= 4) ? atoi(lv) : 3;
  const char *ev = getenv("TFX_RES_EPB");
  for (int lpr = lpr_max; lpr >= 1; --lpr) {
    auto fits = [&](int epb) {
      return lpr == 4 ? res_try<4, W>(h, epb) : lpr == 3 ? res_try<3, W>(h, epb) : lpr == 2 ? res_try<2, W>(h, epb) : res_try<1, W>(h, epb);
    };
    // (three rounds of one-env workgroups or more: two lanes per road with three envs per workgroup stays ahead - 363 us
    // per 10-tick call at cfg1 x 4096 against 392 for the mixed form, whose three envs no longer fit one workgroup)
    if (lpr == 3 && !ev && !getenv("TFX_RES_LPR") &&
        (long)d.E * (((long)2 * d.R + 4 * (h->cfg.m + h->cfg.n) + 63) / 64) >= (long)3 * h->n_cu * 16 && !res_try<3, W>(h, 3))
      continue;
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
    // (round 4, cfg1 x 4096 - three rounds of one-env workgroups over the chip: three envs per workgroup 360 us per
    // 10-tick call against 417; at 1024 envs - one round - one env per workgroup stays ahead, 106 against 130)
    if (lpr == 2 || lpr == 3) {
      const long waves2 = ((long)2 * d.R + (lpr == 3 ? 4 * (h->cfg.m + h->cfg.n) : 0) + 63) / 64;
      if ((long)d.E * waves2 >= (long)3 * h->n_cu * 16 && !fits(3)) (void)fits(1);
    }
    if (lpr >= 2) return TFX_OK;
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
  TFX_INJECT(h);
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
  a.n_ent = 2 * (h->cfg.m + h->cfg.n);
  a.n_int = d.r - a.n_ent;
  a.cols = h->res_epb * d.R + 32;
  const int grid = (d.E + h->res_epb - 1) / h->res_epb;
  a.own_clock = grid == 1 ? 1 : 0;
  h->step_kernel = "k_res";
  const dim3 g(grid), b(h->res_threads);
  if (h->res_lpr == 4) {
    if (d.w) hipLaunchKernelGGL((k_res<4, true>), g, b, h->res_lds, st, d, a);
    else hipLaunchKernelGGL((k_res<4, false>), g, b, h->res_lds, st, d, a);
  } else if (h->res_lpr == 3) {
    if (d.w) hipLaunchKernelGGL((k_res<3, true>), g, b, h->res_lds, st, d, a);
    else hipLaunchKernelGGL((k_res<3, false>), g, b, h->res_lds, st, d, a);
  } else if (h->res_lpr == 2) {
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

int launch_advance(tfx_handle h, int tidx, hipStream_t st) {
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
  TFX_INJECT(h);
  const bool g = h->greedy;
  if (h->het) {
    if (g) hipLaunchKernelGGL((k_advance<true, true, true>), dim3(h->grid_adv), dim3(256), 0, st, d, tidx);
    else hipLaunchKernelGGL((k_advance<true, true>), dim3(h->grid_adv), dim3(256), 0, st, d, tidx);
  } else if (d.layout == 1) {
    if (g) hipLaunchKernelGGL((k_advance<true, false, true>), dim3(h->grid_adv), dim3(256), 0, st, d, tidx);
    else hipLaunchKernelGGL(k_advance<true>, dim3(h->grid_adv), dim3(256), 0, st, d, tidx);
  } else {
    if (g) hipLaunchKernelGGL((k_advance<false, false, true>), dim3(h->grid_adv), dim3(256), 0, st, d, tidx);
    else hipLaunchKernelGGL(k_advance<false>, dim3(h->grid_adv), dim3(256), 0, st, d, tidx);
  }
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

// Two ticks per pass over the cars (tfx_move_tt.hpp): for calls of two ticks or more on the transposed
// layout whose launches fill the chip (validate mode: the W forms carry the spawn-tick plane; heterogeneous cars: the
// HET forms).  A handle
// that can use them (pairs_usable(h)) runs ALL its single ticks through k_move_tt<false>: the one-tick form that
// reads past the rows a pair may have left empty at the top of a column.
bool pairs_usable(tfx_handle h, int n_ticks = 2) {
  const Dev &d = h->d;
  // (a handle whose envs fit k_res never mixes the two: k_res loads its cars from row 0)
  if (!h->pairs || d.layout != 1 || n_ticks < 2 || h->move_variant != 0 || h->res_epb > 0) return false;
  // measured, vehicle-updates/s with / without: cfg2 x 16 envs (272 tiles) 1.8e10 / 2.5e10 and cfg4 x 1 (260) 2.6e10 /
  // 3.8e10 - there four wavefronts per tile (k_move_ts) finish sooner; cfg4 x 4 (1040) 9.0e10 / 6.6e10, cfg2 x 64
  // (1088) 9.1e10 / 7.1e10, cfg2 x 128 1.6e11 / 1.3e11, cfg4 x 8 2.1e11 / 1.8e11, cfg2 x 256 2.5e11 / 2.1e11
  // Single-archetype cars (with or without the side-word plane): always.  With the tiles' walks split over up to eight wavefronts (k_move_tts) the pairs beat the
  // tick-by-tick kernels at every launch size - us per tick / us per fused 10-tick decision, pairs against tick by tick:
  // cfg2 x 1 env (17 tiles) 14.3 / 207 against 18.5 / 215, x 8 16.1 / 226 against 20.3 / 236, x 32 21.6 / 281 against 34.9 /
  // 374, x 48 21.9 / 288 against 52.2 / 535; cfg4 x 1 env (260 tiles) 18.5 / 245 against 23.5 / 275 (closed loop 24.6
  // against 26.7), x 2 27.2 against 37.0, x 3 30.4 against 41.4.  (Single ticks of such handles: single_tick_ts.)
  // Heterogeneous cars have no segmented pass: from four tiles per CU on.
  const long tiles = (long)d.E * d.G;
  return h->pairs == 2 || !h->het || tiles >= (long)h->n_cu * 4;
}

// A single tick of a handle that runs its calls as pairs: launches small enough for k_move_ts (several wavefronts per
// tile, one tick; it reads past the rows a pair's second tick left empty like the one-tick form of k_move_tt does) take
// it - the one-tick k_move_tt walks a tile with ONE wavefront (a 16x16 env's env.step(): 128-row columns).
bool single_tick_ts(const tfx_handle_s *h) {
  const long tiles = (long)h->d.E * h->d.G;
  const long split_below = (h->d.C - 2 > 64) ? (long)h->n_cu * 9 / 2 : (long)h->n_cu * 2;
  return !h->het && h->move_variant == 0 && h->d.layout == 1 && tiles <= split_below;
}

int edge_grid(tfx_handle h) {
  if (h->grid_edge == 0) {  // every block resident at once: a second, nearly empty round would double the time
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (k_edge<false, false>), 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    if (per_cu > 6) per_cu = 6;  // measured at cfg2, 4 / 5 / 6 / 7 blocks per CU: 0.112 / 0.102 / 0.097 / 0.118 ms
    const long tiles = (long)h->d.E * h->d.G;
    long g = (long)h->n_cu * per_cu;
    if (g > (tiles + 3) / 4) g = (tiles + 3) / 4;
    h->grid_edge = (int)(g < 1 ? 1 : g);
  }
  return h->grid_edge;
}

// Wavefronts per tile of a two-tick pass (0: one, k_move_tt).  Measured (us per tick; S = 0 / 2 / 4 / 8):
//   cfg2 (47-row tiles, every road alike) x 64 envs (1088 tiles) 32.3 / 29.1 / 27.1 / 30.9; x 128 (2176) 38.1 / 35.1 /
//   36.7 / 45.8; x 256 (4352) 46.0 / 51.8 / 55.4 / 74.3; x 512 72.2 / 79.7 / 89.8 / 134.8
//   cfg4 (128-car rings) closed loop - a few long tiles among short ones - x 1 env (260 tiles; k_move_ts: 26.6) - / - /
//   27.6 / 24.6; x 2 (37.0) - / - / 29.3 / 27.2; x 3 (41.4) - / - / 30.7 / 30.4; x 4 58.5 / 41.8 / 31.7 / 33.4; x 8 - /
//   47.9 / 39.6 / 44.7; x 16 63.9 / 55.3 / 52.8 / 65.9; prefilled x 16 (every road full) 74.1 / 73.6 / 75.5 / 90.4
//   fused 10-tick agent decisions, cfg4 prefilled, us, with / without: x 2 envs 325 / 725 (tick by tick: 378), x 3 343 / 721,
//   x 4 423 / 738, x 8 582 / 626, x 16 (four segments) 871 / 825
// Rule: the most segments that keep the launch within the chip's wave slots (6 per SIMD); long rings, whose launch is
// as long as its longest tile's walk, take four up to 10 tiles per CU and two up to 20 even beyond that.
int tt_segments(const tfx_handle_s *h) {
  const long tiles = (long)h->d.E * h->d.G, slots = (long)h->n_cu * 24;
  const bool long_rings = h->d.C - 2 > 64;
  if (tiles * 8 <= slots) return 8;
  if (tiles * 4 <= slots || (long_rings && tiles <= (long)h->n_cu * 10)) return 4;
  if (tiles * 2 <= slots || (long_rings && tiles <= (long)h->n_cu * 20)) return 2;
  return 0;
}

// AGENT: inside an agent step; only_risky: the second tick of the envs k_risk sorted out of a pair
// crec: k_tail follows this (two-tick) pass - the road records go out in their 8-byte form (Dev::crec)
// rsw: ... and a k_tail of the same call came before it - the ring indices come from its road state words (Dev::rsw)
template <bool TWO, bool AGENT = false>
int launch_move_tt(tfx_handle h, int tidx, hipStream_t st, int only_risky = 0, bool crec = false, bool rsw = false) {
  // Grid: 16 workgroups per CU, 6 of them resident at once: later rounds of workgroups even out the end of the launch.
  // Measured at cfg2, round 4 (k_tail with its lighter records), split call, same box, workgroups per CU -> ms per tick /
  // ms per pass on the chip: 6 -> 0.396 / 0.739; 8 -> 0.404 / 0.755; 10 -> 0.387-0.392 / 0.702-0.706; 12 -> 0.382 / 0.713;
  // 14 -> 0.379 / 0.696; 16 -> 0.377-0.380 / 0.695-0.698; 18 -> 0.387 / 0.688; 20 -> 0.400 / 0.695; 24 -> 0.401 / 0.681;
  // one workgroup per four tiles (34 per CU) -> 0.412 / 0.698 (tools/sweep_blocks.sh; another box 10 / 15 / 16 / 17:
  // 0.391-0.393 / 0.386 / 0.388-0.389 / 0.388-0.389, agent decision 4.52-4.54 / 4.45 / 4.43-4.46 / 4.39-4.44 ms).
  // Round 3, with the heavier k_tail behind the pass, had its optimum at 10.  More rounds than ~16 per CU cost the split
  // call more than they give the launch.  (Grids of a whole number of workgroups per CU: an "exactly balanced" 2902
  // instead of 3072 workgroups took 0.756.)
  int &resident = h->grid_tt[(TWO ? 1 : 0) + (AGENT ? 2 : 0)];
  if (resident == 0) {
    int per_cu = 0;
    const auto occ = h->d.het ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_move_tt<TWO, AGENT, true, true>, 256, 0)
                     : h->d.w ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_move_tt<TWO, AGENT, true>, 256, 0)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_move_tt<TWO, AGENT, false>, 256, 0);
    if (occ != hipSuccess || per_cu < 1) per_cu = 4;
    if (per_cu > 6) per_cu = 6;
    resident = h->n_cu * per_cu;
  }
  h->step_kernel = "k_move_tt";
  if (h->size_only) return TFX_OK;
  TFX_INJECT(h);
  long grid = (long)resident * 16 / 6;
  if (const char *pc = getenv("TFX_MOVE_BLOCKS_PER_CU")) grid = atoi(pc) > 0 ? (long)atoi(pc) * h->n_cu : grid;
  const long need = ((long)h->d.E * h->d.G + 3) / 4;
  if (grid > need) grid = need;
  if (grid >= 8) grid -= grid % 8;
  if (grid < 1) grid = 1;
  static const bool want_stagger = !(getenv("TFX_STAGGER") && atoi(getenv("TFX_STAGGER")) == 0);
  const bool stagger = want_stagger && TWO && h->split_first && h->split_half >= 0;
  if (stagger && h->split_half == 1) HIPCHK(hipStreamWaitEvent(st, h->split_stagger, 0));
  // Launches that leave most wave slots empty with one wavefront per tile: the tiles' walks split over 2, 4 or 8
  // wavefronts (tfx_move_tts.hpp, tt_segments; plain cars).  TFX_TT_SEG=0 never, 2 whenever the form
  // exists; TFX_TT_SEGS = 2 / 4 / 8 forces the number.
  if (TWO && !only_risky && !h->d.het && h->tt_seg &&
      (h->tt_seg == 2 || (h->split_half < 0 && tt_segments(h) > 0))) {
    const long tiles_all = (long)h->d.E * h->d.G;
    int S = tt_segments(h);
    if (S == 0) S = 2;  // (forced: TFX_TT_SEG=2)
    if (h->tt_segs == 2 || h->tt_segs == 4 || h->tt_segs == 8) S = h->tt_segs;
    const long groups = S == 2 ? (tiles_all + 1) / 2 : tiles_all;
    long gs = (long)h->n_cu * 16;  // (workgroups stride over the tiles)
    if (gs > groups) gs = groups;
    const dim3 g2((unsigned)gs), b2(S == 2 ? 256 : 64 * S);
#define TFX_TTS_LAUNCH(SEGS)                                                                                   \
    do {                                                                                                       \
      if (h->d.w) {                                                                                                      \
        if (AGENT && crec) hipLaunchKernelGGL((k_move_tts<AGENT, true, false, SEGS, true>), g2, b2, 0, st, h->d, tidx);   \
        else if (AGENT) hipLaunchKernelGGL((k_move_tts<AGENT, false, false, SEGS, true>), g2, b2, 0, st, h->d, tidx);     \
        else if (crec && rsw) hipLaunchKernelGGL((k_move_tts<false, true, true, SEGS, true>), g2, b2, 0, st, h->d, tidx); \
        else if (crec) hipLaunchKernelGGL((k_move_tts<false, true, false, SEGS, true>), g2, b2, 0, st, h->d, tidx);       \
        else hipLaunchKernelGGL((k_move_tts<false, false, false, SEGS, true>), g2, b2, 0, st, h->d, tidx);                \
      } else if (AGENT && crec) hipLaunchKernelGGL((k_move_tts<AGENT, true, false, SEGS>), g2, b2, 0, st, h->d, tidx);    \
      else if (AGENT) hipLaunchKernelGGL((k_move_tts<AGENT, false, false, SEGS>), g2, b2, 0, st, h->d, tidx);             \
      else if (crec && rsw) hipLaunchKernelGGL((k_move_tts<false, true, true, SEGS>), g2, b2, 0, st, h->d, tidx);         \
      else if (crec) hipLaunchKernelGGL((k_move_tts<false, true, false, SEGS>), g2, b2, 0, st, h->d, tidx);               \
      else hipLaunchKernelGGL((k_move_tts<false, false, false, SEGS>), g2, b2, 0, st, h->d, tidx);                        \
    } while (0)
    if (S == 8) TFX_TTS_LAUNCH(8);  // (sixteen: cfg4 x 1 env 25.1 against 24.6 us per tick, prefilled 21.4 against 18.5)
    else if (S == 4) TFX_TTS_LAUNCH(4);
    else TFX_TTS_LAUNCH(2);
#undef TFX_TTS_LAUNCH
    HIPCHK(hipGetLastError());
    h->step_kernel = "k_move_tts";
    if (stagger && h->split_half == 0) HIPCHK(hipEventRecord(h->split_stagger, st));
    if (stagger) h->split_first = false;
    return TFX_OK;
  }
  const dim3 g((unsigned)grid), b(256);
  if (TWO && crec && rsw && !AGENT && !h->d.het) {
    constexpr bool CR = TWO && !AGENT;
    if (h->d.w) hipLaunchKernelGGL((k_move_tt<CR, false, true, false, CR, CR>), g, b, 0, st, h->d, tidx, only_risky);
    else hipLaunchKernelGGL((k_move_tt<CR, false, false, false, CR, CR>), g, b, 0, st, h->d, tidx, only_risky);
  } else if (TWO && crec) {  // (k_tail follows)
    constexpr bool CR = TWO;
    if (h->d.het) hipLaunchKernelGGL((k_move_tt<CR, AGENT, true, true, CR>), g, b, 0, st, h->d, tidx, only_risky);
    else if (h->d.w) hipLaunchKernelGGL((k_move_tt<CR, AGENT, true, false, CR>), g, b, 0, st, h->d, tidx, only_risky);
    else hipLaunchKernelGGL((k_move_tt<CR, AGENT, false, false, CR>), g, b, 0, st, h->d, tidx, only_risky);
  } else if (h->d.het) hipLaunchKernelGGL((k_move_tt<TWO, AGENT, true, true>), g, b, 0, st, h->d, tidx, only_risky);
  else if (h->d.w) hipLaunchKernelGGL((k_move_tt<TWO, AGENT, true>), g, b, 0, st, h->d, tidx, only_risky);
  else hipLaunchKernelGGL((k_move_tt<TWO, AGENT, false>), g, b, 0, st, h->d, tidx, only_risky);
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
  s.lights = s.obs + 2 * d.r;
  s.rewards = d.rewards + L * I;
  s.waiting = d.waiting + L * r;
  s.passed_dst = d.passed_dst + L * I;
  s.done_tick = d.done_tick + L;
  if (d.trip_times) s.trip_times = d.trip_times + L * (size_t)d.trip_cap;
  if (d.n_trips) s.n_trips = d.n_trips + L;
  s.rec = d.rec + L * R;
  s.rec2f = d.rec2f + L * R;
  if (d.rsw) s.rsw = d.rsw + L * R;
  if (d.crec) {
    s.crec = d.crec + L * R;
    s.ovf_cnt = d.ovf_cnt + L * R;
  }
  s.rec2c = d.rec2c + L * R;
  if (d.hb) s.hb = d.hb + L * R;
  s.tailx = d.tailx + L * R;
  if (d.taila) s.taila = d.taila + L * R;
  if (d.spawn_arch) s.spawn_arch = d.spawn_arch + L * (size_t)d.n_entry * (size_t)d.spawn_arch_S;
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
// env, the env's ring words staged in LDS.  Not when the arrivals of t+1 are produced by a launch between the two ticks
// (the Poisson stream tick by tick: agent steps; tfx_step generates them up front), not below one env per CU (a handful
// of big envs - cfg4 - has too few workgroups to offer), and not when an env's words do not fit a workgroup's LDS.
// (The greedy controller decides inside the advance.)
constexpr size_t TAIL_LDS_MAX = (size_t)160 * 1024;
bool tail_usable(tfx_handle h) {
  if (!h->tail || (h->poisson && h->d.spawn_stride == 0) || h->d.layout != 1) return false;
  if (tail_lds_bytes(h->d.R, h->d.I, h->het) > TAIL_LDS_MAX) return false;
  // (the halves of a split call - chosen for the whole range, split_usable - keep k_tail whatever their own size)
  return h->tail == 2 || h->d.E >= h->n_cu || h->split_half >= 0;
}

// Two halves on two streams: where each half still runs pairs with k_tail behind them (one env per CU and half).
// Measured at cfg2, ms per tick, one range / two halves: 4096 envs 0.474 / 0.437, 2048 0.238 / 0.227, 1024
// 0.128 / 0.121, 512 0.0759 / 0.0705.  Not while per-kernel timing is on (tfx_profile times launches that own
// the chip).
bool split_usable(tfx_handle h, int n_ticks) {
  if (!h->split || h->prof || n_ticks < 2 || !pairs_usable(h, n_ticks) || !tail_usable(h)) return false;
  if (h->d.E < 2) return false;
  if (h->split == 2) return true;
  // (round 4, us per tick, two halves / one range: 288 envs 61.7 / 47.6, 320 52.3 / 52.5, 352 63.6 / 52.4, 384 56.8 / 62.3,
  // 448 60.9 / 66.3, 512 64.3 / 71.4, 768 84.8 / 93.4: from three quarters of an env per CU and half on)
  return h->d.E / 2 >= h->n_cu * 3 / 4 && (long)(h->d.E / 2) * h->d.G >= (long)h->n_cu * 4;
}

// Workgroup size (the kernel takes any multiple of 64).  Alone on the chip 256 lanes and as many workgroups as fit; as
// one half of a split tfx_step call, next to the other half's pass, 128 lanes (they fit the gaps the pass leaves).
// Round 4 sweep with the staged kernel, cfg2, ms per tick (64 / 128 / 192 / 256 lanes in the halves): 0.419 / 0.399 /
// 0.406 / 0.414; workgroups per CU capped at 2 / 3 / as many as fit: 0.414 / 0.419 / 0.399.
template <bool GREEDY, bool AGENT, bool W, bool HET>
int launch_tail_as(tfx_handle h, int tidx, hipStream_t st, int flags) {
  auto kern = k_tail<GREEDY, AGENT, W, HET>;
  const size_t lds = tail_lds_bytes(h->d.R, h->d.I, HET);
  // (asked of HIP it would fail - and leave its error behind for the next launch's hipGetLastError to find)
  if (lds > TAIL_LDS_MAX) return fail(TFX_ESTATE, "an env's ring words (%zu bytes) do not fit a workgroup's LDS", lds);
  // The attribute belongs to the FUNCTION, not to the handle: only ever raised (see res_try)
  static size_t granted = 64 * 1024;
  if (lds > granted) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    granted = lds;
  }
  if (h->grid_tail == 0) {
    for (int half = 0; half < 2; ++half) {
      const int threads = half ? 128 : 256;
      int per_cu = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds) != hipSuccess || per_cu < 1) per_cu = 1;
      long g = (long)h->n_cu * per_cu;
      if (g > h->d.E) g = h->d.E;
      (half ? h->grid_tail_half : h->grid_tail) = (int)(g < 1 ? 1 : g);
      (half ? h->tail_threads_half : h->tail_threads) = threads;
    }
  }
  if (h->size_only) return TFX_OK;
  TFX_INJECT(h);
  // (inside agent steps the whole-size workgroups win in the halves as well: measured at cfg2, 128 / 192 / 256 lanes:
  // plain calls 0.399 / 0.406 / 0.414 ms per tick, fused decisions 4.72 / 4.63 / 4.51 ms)
  // (... and so do halves of fewer than two envs per CU: cfg2 x 512 envs 65.1 against 72.5 us per tick, x 1024 112.8 /
  // 112.1, x 2048 204.9 / 196.1)
  const bool halves = h->split_half >= 0 && !AGENT && h->d.E >= 2 * h->n_cu;
  const dim3 g(halves ? h->grid_tail_half : h->grid_tail), b(halves ? h->tail_threads_half : h->tail_threads);
  hipLaunchKernelGGL(kern, g, b, lds, st, h->d, tidx, flags);
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

int launch_tail(tfx_handle h, int tidx, hipStream_t st, bool agent = false, int flags = TAIL_LAST) {
  const int sel = (h->greedy ? 1 : 0) | (agent ? 2 : 0) | (h->d.w ? 4 : 0) | (h->d.het ? 8 : 0);
  switch (sel) {
    case 12: return launch_tail_as<false, false, true, true>(h, tidx, st, flags);
    case 13: return launch_tail_as<true, false, true, true>(h, tidx, st, flags);
    case 14: return launch_tail_as<false, true, true, true>(h, tidx, st, flags);
    case 15: return launch_tail_as<true, true, true, true>(h, tidx, st, flags);
    case 0: return launch_tail_as<false, false, false, false>(h, tidx, st, flags);
    case 1: return launch_tail_as<true, false, false, false>(h, tidx, st, flags);
    case 2: return launch_tail_as<false, true, false, false>(h, tidx, st, flags);
    case 3: return launch_tail_as<true, true, false, false>(h, tidx, st, flags);
    case 4: return launch_tail_as<false, false, true, false>(h, tidx, st, flags);
    case 5: return launch_tail_as<true, false, true, false>(h, tidx, st, flags);
    case 6: return launch_tail_as<false, true, true, false>(h, tidx, st, flags);
    default: return launch_tail_as<true, true, true, false>(h, tidx, st, flags);
  }
}

template <bool AGENT>
int launch_edge(tfx_handle h, int tidx, hipStream_t st) {
  TFX_INJECT(h);
  if (h->d.het) hipLaunchKernelGGL((k_edge<AGENT, true, true>), dim3(edge_grid(h)), dim3(256), 0, st, h->d, tidx);
  else if (h->d.w) hipLaunchKernelGGL((k_edge<AGENT, true>), dim3(edge_grid(h)), dim3(256), 0, st, h->d, tidx);
  else hipLaunchKernelGGL((k_edge<AGENT, false>), dim3(edge_grid(h)), dim3(256), 0, st, h->d, tidx);
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

int launch_risk(tfx_handle h, int tidx, hipStream_t st) {
  TFX_INJECT(h);
  hipLaunchKernelGGL(k_risk, dim3(edge_grid(h)), dim3(256), 0, st, h->d, tidx);
  HIPCHK(hipGetLastError());
  return TFX_OK;
}

}  // namespace
