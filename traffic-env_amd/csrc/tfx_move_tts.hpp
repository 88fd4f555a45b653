// tfx_move_tts.hpp - k_move_tts: the two-tick pass (tfx_move_tt.hpp) with every tile's walk split over S = 2, 4 or 8
// wavefronts, for launches that cannot fill the chip's wave slots with one wavefront per tile (a few big envs: cfg4 x 16
// envs is 4160 tiles for 6144 slots, one 16x16 env 17 tiles) - such a launch is as long as its longest tile's serial walk
// (cfg4: 128 rows x two ticks), however many slots idle beside it.
//
// Segment s takes the cars s kseg .. (s + 1) kseg - 1 of every road of the tile (kseg: the tile's longest road over S, a
// multiple of the prefetch depth; cars count old rows first, then this tick's arrivals).  What a segment behind the first
// ("B"; the first is "A") needs to start in the middle of a column is little, because the update is Jacobi:
//   * the pops of tick t in front of it - the `while` of :123 is a prefix: B re-evaluates the head cars until one stays
//     (usually the first), as k_move_ts's segments do;
//   * the tick-t state of the car in front of its first, h-1 (for car h's second tick): one IDM step on the old rows h-1
//     and h-2.
// Every segment takes its own last car through its second tick itself (it holds y(h-1), y(h-2)); the next one stores z
// from its first car on.
// Stores: a segment's land in rows it has already read - except B's first few survivors, which belong in rows below
// h + hb: rows the segment in front may still be reading (the column is compacted by the pops of tick t and by the empty
// rows hb the last pair left on top).  Those are held in registers (at most TFX_KP + 3) until a workgroup barrier behind
// the walks.  Before the walks a barrier separates every read of the road's words and of the start rows from the first store.
// Counts, pops, tail and record meet in LDS; the first segment's lanes write the road's outputs exactly as move_tt_tile does.
// Single-archetype cars, with or without the side-word plane (tfx_step calls and agent steps); bit-identical to k_move_tt - every parity test of the pairs and of the
// agent steps' pairs runs through this kernel as well (TFX_TT_SEG=2, TFX_TT_SEGS=2/4/8), and the fuzzers draw it.
#pragma once
#include "tfx_move_tt.hpp"

namespace tfx {

constexpr int TTS_HOLD = KP + 3;

struct TtsShare {  // per tile of the workgroup and per segment behind the first: what it hands segment A
  int n_wait[64], n_det[64], n_wait1[64], n_det1[64], kpop[64], flags[64];  // flags: 1 = the segment had cars, 2 = far
  float y1x[64], y1v[64], tail_z[64];
};

// the S wavefronts (seg 0 = A, the others like B) of one tile; `active`: the tile exists (every wavefront reaches every
// barrier anyway).  sh: S - 1 records, one per segment behind the first.
// AGENT: inside an agent step - tiles of envs that stand still are skipped (`skip`: also the tiles k_tail takes through
// both ticks itself), `two` = false: the tile of an env k_risk sorted out takes the one-tick form (as in move_tt_tile)
// W: the spawn-tick plane travels with the cars (validate mode): a car's side word is stored wherever its (x, v) is
template <bool AGENT, bool CREC, bool RSW, int S, bool W = false>
__device__ __forceinline__ int move_tt_tile_seg(const Dev &d, const long tile, const bool active, const int lane, const int seg,
                                                const int tick, const int tick_sp, const int tidx, TtsShare *sh,
                                                const bool two = true, const bool skip = false) {
  constexpr int P = TT_P;
  const int C = d.C;
  const int env = active ? (int)(tile / d.G) : 0;
  const int e_slot = active ? d.slot_road[(int)(tile - (long)env * d.G) * 64 + lane] : -1;
  const bool valid = e_slot >= 0;
  const int e = valid ? e_slot : 0;
  const int id = env * d.R + e;
  const int hb0 = (valid && !RSW) ? d.hb[id] : 0;
  const bool run = valid && !skip && !(AGENT && env_frozen(d, env, tick));
  // (nobody stores a road word before the barrier below: the other segment may still have to read it)
  const RoadPrep p = prep_road<RSW>(d, id, env, e, tick, tick_sp, tidx, run, false);
  const int hb = run ? (RSW ? p.hb : hb0) : 0;
  const int n_old = run ? p.n_old : 0;
  const int n_tot = run ? p.n_tot : 0;

  float2 *col = d.xv + ((size_t)(active ? tile : 0) * d.trows) * 64 + lane;
  const float2 *colr = col + (size_t)hb * 64;
  float2 *ocol = d.outb + ((size_t)(active ? tile : 0) * KP) * 64 + lane;
  float *wcol = W ? d.w + ((size_t)(active ? tile : 0) * d.trows) * 64 + lane : nullptr;  // side words: same rows as col
  const float *wcolr = W ? wcol + (size_t)hb * 64 : nullptr;
  float *owcol = W ? d.outw + ((size_t)(active ? tile : 0) * KP) * 64 + lane : nullptr;

  int kmax = n_tot;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int o = __shfl_xor(kmax, off, 64);
    kmax = o > kmax ? o : kmax;
  }
  kmax = __builtin_amdgcn_readfirstlane(kmax);
  // the split: segment s takes cars [s * kseg, (s + 1) * kseg) - kseg a multiple of the prefetch depth, at least 2 P
  // (a later segment needs the two cars in front of its first); short tiles leave the later segments idle
  int kseg = ((kmax + S - 1) / S + P - 1) / P * P;
  if (kseg < 2 * P) kseg = 2 * P;
  const int h = seg * kseg;  // this segment's first car
  const int k_lo = h < kmax ? h : kmax;
  const int k_hi = (h + kseg < kmax) ? h + kseg : kmax;

  auto ld2 = [&](const float2 *ptr) {
    const f2v t = __builtin_nontemporal_load(reinterpret_cast<const f2v *>(ptr));
    return make_float2(t.x, t.y);
  };
  // old state of car k: a row of the column, or a car spawned this tick queueing behind the tail (:97-114)
  auto old_car = [&](int k) {
    if (k < n_old) return ld2(&colr[(size_t)k * 64]);
    return make_float2(spawned_x(d, p.xs0, k - n_old), d.car_v);
  };
  auto old_w = [&](int k) { return (W && k < n_old) ? wcolr[(size_t)k * 64] : (float)tick; };  // spawned this tick: w = tick
  auto idm = [&](float x, float v, float xl, float vl, float ll, float &xn, float &vn) {
    const bool off_domain = __builtin_amdgcn_ballot_w64(!idm_fast_domain(v)) != 0ull;
    if (d.fastdiv && !off_domain) idm_step_fast(d, x, v, xl, vl, ll, xn, vn);
    else idm_step(d, x, v, xl, vl, ll, xn, vn);
  };

  float xprev = p.xL, vprev = 0.0f, llv = 0.0f;  // OLD state of the car ahead (Jacobi); starts as the fake leader
  float y1x = 0.0f, y1v = 0.0f, y2x = 0.0f, y2v = 0.0f;  // NEW states of cars k-1 and k-2
  float y1w = 0.0f;                                      // side word of car k-1
  int kpop = 0, n_wait = 0, n_det = 0, n_wait1 = 0, n_det1 = 0;
  bool open = true, far = false;
  bool pend = false, pend_int = false;
  int kq1 = 0x7fffffff;
  float tail_z = 0.0f;
  const int kq = (p.ld > p.lc) ? C - 1 - p.ld : 0x7fffffff;
  const bool mine = n_tot > k_lo;  // this lane's road has cars in this segment's range

  if (seg > 0) {
    // ---- B's start: the pops of tick t in front of car h, then the old and new state of car h-1 -----------------
    for (int k = 0;; ++k) {
      const bool act = mine && open && k < h;
      if (__builtin_amdgcn_ballot_w64(act) == 0ull) break;
      if (act) {
        const float2 c = old_car(k);
        float xn, vn;
        idm(c.x, c.y, xprev, vprev, llv, xn, vn);
        open = xn > d.length;
        kpop += open ? 1 : 0;
        xprev = c.x;
        vprev = c.y;
        llv = d.car_l;
      }
    }
    if (__builtin_amdgcn_ballot_w64(mine) != 0ull) {
      float2 c1 = make_float2(0.0f, 0.0f), c2 = make_float2(0.0f, 0.0f);
      if (mine) {  // (h >= 2 P: cars h-1 and h-2 exist)
        c1 = old_car(h - 1);
        c2 = old_car(h - 2);
      }
      float xn = 0.0f, vn = 0.0f;
      // (car h-2's leader plays no part: only car h-1's new state is needed; while `open` holds car h-1 popped and car h
      // is the road's new head - its second tick is the edge work's, no leader state is read)
      idm(c1.x, c1.y, c2.x, c2.y, d.car_l, xn, vn);
      if (mine) {
        xprev = c1.x;
        vprev = c1.y;
        llv = d.car_l;
        y1x = xn;
        y1v = vn;
        if (W) y1w = old_w(h - 1);
        if (!open) kq1 = C - 1 - ring_adv(p.ld, kpop, C) + kpop;  // (set by the first survivor, which sits in A's range)
      }
    }
  }
  __syncthreads();  // every read of the road's words and of B's start rows has happened: stores may begin
  if (seg == 0 && run) {
    if (!RSW && hb) d.hb[id] = 0;  // (this walk writes the column compacted)
    if (!RSW && n_tot != n_old) d.lastcar[id] = p.lc;
  }

  // where the next surviving car goes: one row further down per survivor (compacted by the pops in front of it)
  float2 *wp = col + (size_t)(seg ? (kpop > KP ? k_lo : k_lo - kpop) : 0) * 64;
  const float2 *const hold_below = col + (size_t)(k_lo + hb) * 64;  // B: rows the segment in front may still be reading
  // (B's stores go to consecutive rows, one per car from its first stored car on: the held ones are rows
  // hold_first, hold_first + 1, ...)
  float2 hold[TTS_HOLD];
  float hold_w[W ? TTS_HOLD : 1];
  float2 *hold_first = nullptr;
  int n_hold = 0;
  // (x, v) - and, W, the side word sw - of a car into row `ptr` of the column
  auto st2 = [&](float2 *ptr, float a, float b, float sw) {
    if (seg > 0 && ptr < hold_below && n_hold < TTS_HOLD) {
      // (unrolled select chain instead of an indexed store: the array stays in registers)
#pragma unroll
      for (int q = 0; q < TTS_HOLD; ++q)
        if (q == n_hold) {
          hold[q] = make_float2(a, b);
          if (W) hold_w[q] = sw;
        }
      if (n_hold == 0) hold_first = ptr;
      ++n_hold;
      return;
    }
    f2v t;
    t.x = a;
    t.y = b;
    __builtin_nontemporal_store(t, reinterpret_cast<f2v *>(ptr));
    if (W) wcol[ptr - col] = sw;
  };

  // Car k through tick t, car k-1 through tick t+1 (move_tt_tile's step, plain cars, always two ticks).
  // mode 0: a car; 1: the road's last car's second tick only; 2: A's last car's second tick only (B continues the road)
  auto step = [&](int k, float x, float v, float sw, int mode) {
    float xn = 0.0f, vn = 0.0f, zx = 0.0f, zv = 0.0f;
    const bool bad = (mode == 0 && !idm_fast_domain(v)) || !idm_fast_domain(y1v);
    const bool off_domain = __builtin_amdgcn_ballot_w64(bad) != 0ull;
    if (d.fastdiv && !off_domain) {
      if (mode == 0) idm_step_fast(d, x, v, xprev, vprev, llv, xn, vn);
      idm_step_fast(d, y1x, y1v, y2x, y2v, d.car_l, zx, zv);
    } else {
      if (mode == 0) idm_step(d, x, v, xprev, vprev, llv, xn, vn);
      idm_step(d, y1x, y1v, y2x, y2v, d.car_l, zx, zv);
    }
    if (two && pend) {  // car k-1: the new head keeps its tick-t state (the edge work moves it), the others are a tick ahead
      st2(wp, pend_int ? zx : y1x, pend_int ? zv : y1v, y1w);
      wp += 64;
      if (pend_int) {
        const float wq1 = (k - 1 >= kq1) ? zx : zv;
        n_wait1 += (wq1 < d.thresh) ? 1 : 0;
        n_det1 += (zx > d.near_end) ? 1 : 0;
        if (mode == 1) tail_z = zx;
      }
    }
    if (mode != 0) return;
    xprev = x;
    vprev = v;
    llv = d.car_l;
    const bool was_open = open;
    const bool pop = open && (xn > d.length);  // the while loop of :123
    open = pop;
    if (pop) {
      if (kpop < KP) {
        ocol[(size_t)kpop * 64] = make_float2(xn, vn);
        if (W) owcol[(size_t)kpop * 64] = sw;
      } else {  // third pop: no survivor has been written yet - from here on every car stays in its row
        st2(&col[(size_t)k * 64], xn, vn, sw);
        wp = col + (size_t)(k + 1) * 64;
      }
      far = far || ((xn - d.length) > d.length);
      ++kpop;
    } else if (two) {
      pend = true;
      pend_int = !was_open;
      if (was_open) kq1 = C - 1 - ring_adv(p.ld, kpop, C) + kpop;  // (kpop is final: this is the first survivor)
    } else {  // (the one-tick form: the survivor goes straight to its row)
      st2(wp, xn, vn, sw);
      wp += 64;
    }
    const float wq = (k >= kq) ? xn : vn;
    n_wait += (wq < d.thresh) ? 1 : 0;
    n_det += (xn > d.near_end) ? 1 : 0;
    y2x = y1x;
    y2v = y1v;
    y1x = xn;
    y1v = vn;
    if (W) y1w = sw;
  };

  // ---- this segment's cars in memory: rows k_lo .. min(k_hi, n_old) - 1 of the live part, P rows in flight ---------
  float2 pf[P];
  float pfw[W ? P : 1];
#pragma unroll
  for (int u = 0; u < P; ++u) {
    const bool in = k_lo + u < n_old && k_lo + u < k_hi;
    pf[u] = in ? ld2(&colr[(size_t)(k_lo + u) * 64]) : make_float2(0.0f, 0.0f);
    if (W) pfw[u] = in ? wcolr[(size_t)(k_lo + u) * 64] : 0.0f;
  }
  for (int k0 = k_lo; k0 < k_hi; k0 += P) {
#pragma unroll
    for (int u = 0; u < P; ++u) {
      const int k = k0 + u;
      if (k < k_hi) {
        const float2 cur = pf[u];
        const float curw = W ? pfw[u] : 0.0f;
        if (k + P < k_hi) {
          pf[u] = (k + P < n_old) ? ld2(&colr[(size_t)(k + P) * 64]) : make_float2(0.0f, 0.0f);
          if (W) pfw[u] = (k + P < n_old) ? wcolr[(size_t)(k + P) * 64] : 0.0f;
        }
        if (k < n_old) step(k, cur.x, cur.y, curw, 0);
        else if (k < n_tot) step(k, spawned_x(d, p.xs0, k - n_old), d.car_v, (float)tick, 0);  // arrivals queue behind the tail (:97-114)
      }
    }
  }
  // ---- the second tick of the segment's last car ------------------------------------------------------------------
  {
    const bool road_ends_here = n_tot > k_lo && n_tot <= k_hi;       // the road's last car is this segment's
    const bool hand_over = n_tot > k_hi && k_hi > k_lo;              // the next segment continues the road behind this one's last car
    if (__builtin_amdgcn_ballot_w64(pend && (road_ends_here || hand_over)) != 0ull) {
      if (pend && road_ends_here) step(n_tot, 0.0f, 0.0f, 0.0f, 1);
      else if (pend && hand_over) step(k_hi, 0.0f, 0.0f, 0.0f, 2);
    }
  }

  // ---- the segments meet ------------------------------------------------------------------------------------------
  if (seg > 0) {
    TtsShare &o = sh[seg - 1];
    o.n_wait[lane] = n_wait;
    o.n_det[lane] = n_det;
    o.n_wait1[lane] = n_wait1;
    o.n_det1[lane] = n_det1;
    o.kpop[lane] = kpop;
    o.flags[lane] = (mine ? 1 : 0) | (far ? 2 : 0);
    o.y1x[lane] = y1x;
    o.y1v[lane] = y1v;
    o.tail_z[lane] = tail_z;
  }
  __syncthreads();  // every segment has read every row it needed; the later segments' counts are in LDS
  if (seg > 0) {
#pragma unroll
    for (int q = 0; q < TTS_HOLD; ++q)
      if (q < n_hold) {
        f2v t;
        t.x = hold[q].x;
        t.y = hold[q].y;
        __builtin_nontemporal_store(t, reinterpret_cast<f2v *>(hold_first + (size_t)q * 64));
        if (W) wcol[(hold_first - col) + (size_t)q * 64] = hold_w[q];
      }
    return 0;
  }
  if (!run) return 0;
#pragma unroll
  for (int q = 0; q < S - 1; ++q) {
    if ((q + 1) * kseg < kmax) {  // (segment q + 1 had a range)
      const TtsShare &o = sh[q];
      const int fl = o.flags[lane];
      n_wait += o.n_wait[lane];
      n_det += o.n_det[lane];
      n_wait1 += o.n_wait1[lane];
      n_det1 += o.n_det1[lane];
      far = far || (fl & 2);
      if (fl & 1) {  // it saw cars of this road: the last such segment's pop count is the road's, the tail is its
        kpop = o.kpop[lane];
        y1x = o.y1x[lane];
        y1v = o.y1v[lane];
        tail_z = o.tail_z[lane];
      }
    }
  }
  // ---- phase W (move_tt_tile's, plain cars) -------------------------------------------------------------------------
  if (e < d.r) {
    int *ob = d.obs + (size_t)env * d.obs_len;
    if (n_tot > 0 && !two) {  // (a pair: edge_tile adds both ticks' waiting counts at once and stores `detected`)
      d.waiting[(size_t)env * d.r + e] += n_wait;
      ob[d.r + e] = n_det;
    }
    if (AGENT && !two) ob[e] = (tidx > 0) ? ob[e] + kpop : kpop;  // accumulates over the agent step
    else if (!two) ob[e] = kpop;
    if (kpop > 0) d.passed_dst[(size_t)env * d.I + e % d.I] = 1;
  }
  const float tail_x = y1x;  // new x of the last car processed (0 if there was none)
  if (CREC) {
    d.crec[id] = make_int2(crec_pack<false>(kpop, n_tot, p.ld, p.lc, 0, kpop > KP, p.ovf_sp > 0), __float_as_int(tail_x));
    if (p.ovf_sp > 0) d.ovf_cnt[id] = p.ovf_sp;
  } else {
    d.rec[id] = make_int4(rec_pack(kpop, p.ld, C), rec_y(p.ovf_sp, kpop > KP), __float_as_int(tail_x), n_tot);
  }
  if (two) {
    d.rec2f[id] = make_float2(y1v, tail_z);
    d.rec2c[id] = rec2c_pack(n_wait + n_wait1, n_det1, n_det, n_tot > 0);
  }
  if (far || kpop > KP) d.env_flag[env] = tick + 1;
  if (!two) d.leadx[id] = p.xL;  // (read by tfx_export_ring only: the second tick of a pair writes its own)
  return n_tot;
}

// a workgroup = TPW tiles x S segments (S = 2: two tiles; 4, 8: one)
template <bool AGENT, bool CREC, bool RSW, int S, bool W = false>
__global__ __launch_bounds__(S == 2 ? 256 : 64 * S)
__attribute__((amdgpu_waves_per_eu(S == 2 ? (W ? TT_WAVES_W : TT_WAVES) : 4, S == 2 ? (W ? TT_WAVES_W : TT_WAVES) : 4)))
void k_move_tts(const Dev d, const int tidx) {
  static_assert(!RSW || (CREC && !AGENT), "road state words: between the pairs of a plain tfx_step call");
  constexpr int TPW = S == 2 ? 2 : 1;
  __shared__ TtsShare s_share[TPW][S - 1];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int seg = wv % S, tl = wv / S;
  const int tick = *d.tickA;
  const long tiles = (long)d.E * d.G;
  const long n_groups = (tiles + TPW - 1) / TPW;
  const int tick_sp = (d.spawn_mode == TFX_SPAWN_PERIODIC) ? tick % d.spawn_period : 0;
  unsigned long long my_updates = 0;
  for (long pr = blockIdx.x; pr < n_groups; pr += gridDim.x) {
    const long tile = pr * TPW + tl;
    const bool active = tile < tiles;
    // (k_move_tt's rules for the envs k_risk sorted out of this pair: behind k_tail their tiles are left alone, otherwise
    // they take the one-tick form here and a restricted launch brings the second tick)
    const int env = active ? (int)(tile / d.G) : 0;
    const bool sorted_out = AGENT && active && risk_word(d, env, tidx) == tick + 1;
    const bool two = CREC || !sorted_out;
    if (AGENT && active && !two && lane == 0) risk_any_word(d, tidx) = tick + 1;
    my_updates += (unsigned long long)move_tt_tile_seg<AGENT, CREC, RSW, S, W>(d, tile, active, lane, seg, tick, tick_sp, tidx, s_share[tl], two,
                                                                          CREC && sorted_out);
    __syncthreads();  // (the LDS words are free for the next tiles)
  }
  for (int off = 32; off > 0; off >>= 1) my_updates += __shfl_down(my_updates, off);
  if (lane == 0 && my_updates) veh_add(d.veh, my_updates);
  if (blockIdx.x == 0 && threadIdx.x == 0) *d.tickB = tick;
}

}  // namespace tfx
