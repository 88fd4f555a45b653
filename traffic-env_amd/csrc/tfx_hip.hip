// tfx_hip.hip - MI355X (gfx950 / CDNA4) implementation of the IDM traffic-env tick behind the
// C ABI of include/tfx.h.  Written for wave64; no other target is supported.
//
// One tick (reference: gym_traffic/envs/traffic_env.py:224-248, TrafficEnv._step) is
//   * for envs that fit a compute unit's LDS: part of ONE launch per tfx_step / tfx_agent_step call -
//     k_res (tfx_resident.hpp), the cars resident on chip for all the ticks of the call;
//   * otherwise per-tick kernels, from two ticks on as PAIRS:
//       k_move_tt (tfx_move_tt.hpp)   the cars through two ticks per trip through HBM
//       k_tail (tfx_tail.hpp)         advance of tick t, the deferred cars' tick t+1, advance of t+1: a workgroup per env
//     with the env range in two halves on two streams (tfx_sequence.hpp; validate mode: the W forms of both kernels,
//     heterogeneous cars: the HET forms); small launches and the ring layout: k_move_t / k_move_ts / k_move_dma /
//     k_move<WPR> + k_advance.
// This file is the C ABI itself; the handle is in tfx_handle.hpp, kernel choice and launch geometry in tfx_launch.hpp,
// the launch sequences of tfx_step / tfx_agent_step in tfx_sequence.hpp, the cold kernels in tfx_misc.hpp.
#include <cmath>
#include <new>

#include "tfx_sequence.hpp"

namespace {

// Everything a captured launch sequence bakes into its kernel arguments (a stale graph is never replayed, even when a
// re-allocated buffer lands on the address the old one had: input_gen)
void graph_key(tfx_handle h, char *key, size_t n, int n_ticks, int remi, const void *aobs, const void *areward,
               const void *adone) {
  const Dev &d = h->d;
  snprintf(key, n, "%llu|%u|%u|%d.%d.%d.%d|%d%d%d|%ld|%d|%d|%p|%p|%p|%p|%d|%d|%ld|%p|%d|%d|%p|%p|%p|%p|%p|%p|%p|%p",
           h->input_gen, h->ps.seed_lo, h->ps.seed_hi, h->ps.n_cdf, h->ps.regular, h->ps.every, h->ps.burst, (int)h->poisson,
           (int)h->greedy, h->greedy_spacing, d.spawn_stride, n_ticks, remi, aobs, areward, adone, (const void *)d.action,
           d.action_mode, d.action_period, d.action_stride, (const void *)d.spawn, d.spawn_mode, d.spawn_period, (void *)d.xv,
           (void *)d.w, (void *)d.obs, (void *)d.rewards, (void *)d.leading, (void *)d.lastcar, (void *)d.waiting,
           (void *)d.done_tick);
}

}  // namespace

extern "C" int tfx_agent_step(tfx_handle h, int32_t n_ticks, int32_t remi, float *aobs, float *areward,
                              uint8_t *adone, void *stream) {
  if (int rc = check_handle(h, true)) return rc;
  if (n_ticks < 1) return fail(TFX_EINVAL, "n_ticks < 1");
  if (h->action_per_tick)
    return fail(TFX_EINVAL, "the fused agent step holds ONE action for all its ticks (bind the action buffer "
                            "with per_tick = 0)");
  hipStream_t st = (hipStream_t)stream;
  (void)h->d;
  // a batch whose halves still fill the chip: two halves on two streams (k_tail of one under the pass of the other)
  // (decisions split later than plain calls: fused 10-tick decisions at cfg2, us, two halves / one range - the captured
  // graph: 384 envs 715 / 687, 448 747 / 726, 512 808 / 796-807, 768 1064 / 1019-1064, 1024 1251 / 1256, 4096 4380 / 4500 -
  // from two envs per CU and half on)
  const bool split = !res_usable(h, n_ticks) && !h->poisson && split_usable(h, n_ticks) &&
                     (h->split == 2 || h->d.E / 2 >= 2 * h->n_cu);
  if (split) {
    if (int rc = ensure_split(h, st)) return rc;
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
  graph_key(h, key, sizeof key, n_ticks, remi, aobs, areward, adone);
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
      if (tail_usable(h)) (void)launch_tail(h, 0, nullptr, true);
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

namespace {

// the launches of a tfx_step call on the per-tick kernels, on `st`
int step_body(tfx_handle h, int n_ticks, hipStream_t st) {
  if (int rc = launch_greedy(h, st)) return rc;  // (the decision of the call's first tick; later ones: advance_item)
  if (h->poisson && n_ticks > 0) {
    // the arrivals of the whole call (in chunks of the rows the count buffer holds) in ONE launch each: they depend
    // on nothing but the stream, and a launch per tick was the longest one of a cfg4 tick
    int rc = TFX_OK;
    // a chunk's kernels count ticks from 0 (row t of the count buffer is the chunk's tick t); a per-tick ACTION buffer is
    // indexed by the tick of the whole call, so its base moves along with the chunks
    const int *const act0 = h->d.action;
    for (int done = 0; done < n_ticks && rc == TFX_OK;) {
      const int chunk = n_ticks - done < h->poisson_rows ? n_ticks - done : h->poisson_rows;
      rc = launch_poisson(h, chunk, st);
      h->d.spawn_stride = (long)h->d.E * h->d.n_entry;
      if (act0 && h->action_per_tick) h->d.action = act0 + (size_t)done * h->d.action_stride;
      if (rc == TFX_OK) rc = step_chunk(h, chunk, st);
      h->d.spawn_stride = 0;
      h->d.action = act0;
      done += chunk;
    }
    return rc;
  }
  return step_chunk(h, n_ticks, st);
}

// the same as a captured graph: captured once per distinct sequence (graph_key), replayed afterwards
int step_graph(tfx_handle h, int n_ticks, hipStream_t st) {
  char key[640];
  graph_key(h, key, sizeof key, n_ticks, -1, nullptr, nullptr, nullptr);
  if (!h->st_exec || h->st_key != key) {
    if (h->st_exec) { (void)hipGraphExecDestroy(h->st_exec); h->st_exec = nullptr; }
    if (h->st_graph) { (void)hipGraphDestroy(h->st_graph); h->st_graph = nullptr; }
    if (!h->ag_stream) HIPCHK(hipStreamCreateWithFlags(&h->ag_stream, hipStreamNonBlocking));
    // grids are sized outside the capture (occupancy queries, function attributes)
    h->size_only = true;
    (void)launch_advance(h, 0, nullptr);
    if (pairs_usable(h)) {
      (void)launch_move_tt<true>(h, 0, nullptr);
      (void)launch_move_tt<false>(h, 0, nullptr);
      if (single_tick_ts(h)) (void)launch_move(h, 0, nullptr);  // (a call's odd last tick)
      if (tail_usable(h)) (void)launch_tail(h, 0, nullptr);
    } else {
      (void)launch_move(h, 0, nullptr);
    }
    h->size_only = false;
    if (pairs_usable(h)) (void)edge_grid(h);
    const long long pair0 = h->pair_ticks, tail0 = h->tail_ticks;
    HIPCHK(hipStreamBeginCapture(h->ag_stream, hipStreamCaptureModeThreadLocal));
    const int rc = step_body(h, n_ticks, h->ag_stream);
    hipGraph_t g = nullptr;
    const hipError_t ce = hipStreamEndCapture(h->ag_stream, &g);
    h->st_pair = h->pair_ticks - pair0;  // (the capture ran no kernel: every replay counts, see below)
    h->st_tail = h->tail_ticks - tail0;
    h->pair_ticks = pair0;
    h->tail_ticks = tail0;
    if (rc != TFX_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (ce != hipSuccess) return fail(TFX_EDEVICE, "hipStreamEndCapture: %s", hipGetErrorString(ce));
    h->st_graph = g;
    HIPCHK(hipGraphInstantiate(&h->st_exec, g, nullptr, nullptr, 0));
    h->st_key = key;
  }
  HIPCHK(hipGraphLaunch(h->st_exec, st));
  h->pair_ticks += h->st_pair;
  h->tail_ticks += h->st_tail;
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
  if (cfg->layout != 0 && cfg->layout != 1) return fail(TFX_EINVAL, "layout must be 0 (ring) or 1 (transposed)");
  const int n_arch = cfg->n_archetypes <= 1 ? 1 : cfg->n_archetypes;
  if (n_arch > TFX_MAX_ARCH) return fail(TFX_EINVAL, "at most %d archetype rows", TFX_MAX_ARCH);
  bool het = n_arch > 1;
  float arch_rows[TFX_MAX_ARCH][8] = {};
  for (int a = 0; a < n_arch; ++a) {
    const float single[8] = {cfg->car_v, cfg->car_l, cfg->car_a, cfg->car_delta, cfg->car_v0, cfg->car_b, cfg->car_T, cfg->car_s0};
    memcpy(arch_rows[a], cfg->n_archetypes >= 1 ? cfg->arch[a] : single, sizeof single);
    const float delta = arch_rows[a][3];
    if (!(delta > 0.0f) || !(delta <= 64.0f))
      return fail(TFX_EINVAL, "archetype %d: delta = %g - the exponent must be in (0, 64]", a, delta);
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
  if (const char *gv = getenv("TFX_TT_SEG")) h->tt_seg = atoi(gv);
  if (const char *gv = getenv("TFX_TT_SEGS")) h->tt_segs = atoi(gv);
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
  d.risk_a = cfg->car_a;
  if (het)
    for (int a = 0; a < n_arch; ++a) d.risk_a = a == 0 ? arch_rows[0][2] : fmaxf(d.risk_a, arch_rows[a][2]);
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
  const size_t o_rec2 = off;  off = align_up(off + (d.layout == 1 ? ER * sizeof(float2) : 0), 256);
  const size_t o_rec2c = off; off = align_up(off + (d.layout == 1 ? ER * sizeof(int) : 0), 256);
  const size_t o_crec = off;  off = align_up(off + (d.layout == 1 ? ER * sizeof(int2) : 0), 256);
  const size_t o_ovfc = off;  off = align_up(off + (d.layout == 1 ? ER * sizeof(int) : 0), 256);
  const size_t o_rsw = off;   off = align_up(off + (d.layout == 1 ? ER * sizeof(int) : 0), 256);
  const size_t o_tail = off;  off = align_up(off + ER * sizeof(float), 256);
  const size_t o_flag = off;  off = align_up(off + (size_t)d.E * sizeof(int), 256);
  const size_t o_risk = off;  off = align_up(off + 2 * (size_t)d.E * sizeof(int), 256);
  d.trows = d.C - 2;  // (padding the tile stride off the power of two was measured: slightly slower)
  const size_t n_tpairs = (size_t)d.E * d.G * (size_t)d.trows * 64;  // (x, v) pairs of a transposed array
  // outbox: TFX_KP rows per tile (the cars a road hands over in a tick; round 1 kept a T-sized one)
  const size_t n_opairs = (size_t)d.E * d.G * KP * 64;
  const size_t o_outb = off;  off = align_up(off + (d.layout == 1 ? n_opairs * sizeof(float2) : 0), 256);
  const size_t o_outw = off;  off = align_up(off + (d.layout == 1 && cfg->planes == 3 ? n_opairs * sizeof(float) : 0), 256);
  const size_t o_lead = off;  off = align_up(off + ER * sizeof(float), 256);
  const size_t o_taila = off; off = align_up(off + (het ? ER * sizeof(int) : 0), 256);
  const size_t o_hb = off;    off = align_up(off + (d.layout == 1 ? ER : 0), 256);
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
  d.rec2f = (float2 *)(base + o_rec2);
  d.rec2c = (int *)(base + o_rec2c);
  d.crec = d.layout == 1 ? (int2 *)(base + o_crec) : nullptr;
  d.ovf_cnt = d.layout == 1 ? (int *)(base + o_ovfc) : nullptr;
  d.rsw = d.layout == 1 ? (int *)(base + o_rsw) : nullptr;
  d.tailx = (float *)(base + o_tail);
  d.env_flag = (int *)(base + o_flag);
  d.env_risk = (int *)(base + o_risk);
  d.risk_stride = d.E;
  d.outb = (float2 *)(base + o_outb);
  d.outw = (float *)(base + o_outw);
  d.leadx = (float *)(base + o_lead);
  d.hb = d.layout == 1 ? (uint8_t *)(base + o_hb) : nullptr;
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
  d.risk_any = (int *)(base + o_misc + 56);   // two words
  d.slow_pairs = (unsigned long long *)(base + o_misc + 88);
  h->tick2 = (int *)(base + o_misc + 64);     // tickA, tickB, risk_any[2] of the second half
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
  if (h->st_exec) (void)hipGraphExecDestroy(h->st_exec);
  if (h->st_graph) (void)hipGraphDestroy(h->st_graph);
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
  d.lights = b->obs + 2 * d.r; d.lights_stride = d.obs_len;
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
  h->ps.regular = h->ps.every = h->ps.burst = 0;
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

int tfx_set_regular(tfx_handle h, int32_t every, int32_t burst, uint64_t seed) {
  if (int rc = check_handle(h, false)) return rc;
  if (every < 0 || burst < 1) return fail(TFX_EINVAL, "every must be >= 0 and burst >= 1");
  Dev &d = h->d;
  if (d.n_entry < 1) return fail(TFX_EINVAL, "no entry roads");
  ++h->input_gen;
  if (h->dev_ps) { (void)hipFree(h->dev_ps); h->dev_ps = nullptr; }
  long rows = ((long)32 << 20) / ((long)d.E * d.n_entry * 4);
  h->poisson_rows = (int)(rows < 1 ? 1 : (rows > 64 ? 64 : rows));
  const size_t n_counts = (size_t)h->poisson_rows * d.E * d.n_entry;
  const size_t bytes = (n_counts + 2 * (size_t)d.E) * 4;
  HIPCHK(hipMalloc(&h->dev_ps, bytes));
  HIPCHK(hipMemset(h->dev_ps, 0, bytes));  // (tick counters and car indices start at 0)
  int *base = (int *)h->dev_ps;
  h->ps = PoissonDev{};
  h->ps.counts = base;
  h->ps.gap_left = base + n_counts;
  h->ps.draws = (unsigned *)(base + n_counts + d.E);
  h->ps.seed_lo = (unsigned)seed;
  h->ps.seed_hi = (unsigned)(seed >> 32);
  h->ps.regular = 1;
  h->ps.every = every;
  h->ps.burst = burst;
  d.spawn = h->ps.counts;
  d.spawn_stride = 0;
  d.spawn_mode = TFX_SPAWN_COUNTS;
  h->spawn_per_tick = 0;
  h->poisson = true;  // (an arrival stream drawn on the device: every path of tfx_set_poisson serves it)
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
  // Launch-bound handles (a few thousand tiles: cfg4 x 16, a 16x16 grid x 256 envs) replay the call's launches as a HIP
  // graph, like the fused agent step does: at cfg4 x 16 the kernels of a tick add up to 59 us of its 65.6
  // (profiles/r04_cfg4_closed_loop_trace.txt).  Not while kernels are timed, not for calls that split over two streams.
  const Dev &d = h->d;
  if (h->use_graph && !h->prof && n_ticks >= 4 && d.layout == 1 && (long)d.E * d.G <= (long)h->n_cu * 24 &&
      !split_usable(h, n_ticks))
    return step_graph(h, n_ticks, st);
  return step_body(h, n_ticks, st);
}

}  // extern "C"

extern "C" {

int tfx_move_cars(tfx_handle h, void *stream) {
  if (int rc = check_handle(h, true)) return rc;
  if (int rc = launch_greedy(h, (hipStream_t)stream)) return rc;
  if (int rc = launch_inputs(h, (hipStream_t)stream)) return rc;
  return (pairs_usable(h) && !single_tick_ts(h)) ? launch_move_tt<false>(h, 0, (hipStream_t)stream) : launch_move(h, 0, (hipStream_t)stream);
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

int tfx_slow_pairs(tfx_handle h, uint64_t *pairs, void *stream) {
  if (int rc = check_handle(h, false)) return rc;
  if (!pairs) return fail(TFX_EINVAL, "pairs is null");
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  unsigned long long v = 0;
  HIPCHK(hipMemcpy(&v, h->d.slow_pairs, sizeof v, hipMemcpyDeviceToHost));
  *pairs = (uint64_t)v;
  return TFX_OK;
}

int tfx_split_ticks(tfx_handle h, int64_t *ticks) {
  if (int rc = check_handle(h, false)) return rc;
  if (ticks) *ticks = h->split_ticks;
  return TFX_OK;
}

const char *tfx_step_kernel(tfx_handle h) { return h ? h->step_kernel : ""; }

int tfx_debug_fail_after(tfx_handle h, int32_t n_launches) {
  if (int rc = check_handle(h, false)) return rc;
  h->fail_after = n_launches > 0 ? n_launches : 0;
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

// host-only: replay of the reference's seeded arrival generators for many envs
#include "tfx_arrivals.cpp"
