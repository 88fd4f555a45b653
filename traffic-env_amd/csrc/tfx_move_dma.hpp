// tfx_move_dma.hpp - k_move_dma: the HBM-facing move kernel for rings of at most 64 cars.
//
// A wavefront owns a TILE of 64 consecutive roads and works in three wave-local phases (no block
// barrier; the four waves of a block are independent):
//   M  lane j prepares road j of the tile (prep_road): ring indices, light state -> fake-leader x,
//      spawn pushes.  All per-road scalar work happens 64 roads at a time with coalesced loads.
//      The lane keeps the result in registers; phase C fetches it with v_readlane, so per-road
//      values are SGPRs there.
//   C  sub-tiles of S consecutive roads are one contiguous 16-byte-aligned span of S*C (x, v)
//      pairs, copied to LDS by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB per
//      instruction, no VGPR holds the bytes in flight).  With NBUF = 2 the next sub-tile is in
//      flight while the current one is computed (counted s_waitcnt vmcnt).  Per road, lane k reads
//      car k's (x, v) from the image by ring slot (one ds_read_b64), finds its leader either one
//      ring slot ahead in the image (LEADER_LDS: the +1 LDS access) or by a DPP wave shift that
//      injects the fake leader into lane 0, runs the IDM and writes the new (x, v) back into the
//      image.  The four per-road predicates (waiting, detected, crossed the road end, far beyond
//      it) become four 64-bit ballots that go to lane j with v_writelane - no scalar counting in
//      the loop.  The finished sub-tile is written back to HBM as it came, 16 B per lane.
//   W  lane j turns road j's ballots into waiting / detected / passed counts, the pop prefix and
//      the handoff record, and writes them coalesced.
// CC > 0 fixes the ring capacity at compile time (immediate offsets, constexpr DMA counts);
// CC = 0 reads it from the config (single-buffered only).  Needs an even capacity (16-byte road
// records); otherwise the generic k_move<1> is used.
#pragma once
#include <type_traits>

#include "tfx_common.hpp"

namespace tfx {

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

// v_writelane_b32: put a wave-uniform value into one lane of a VGPR (no clang builtin on this
// toolchain; the LLVM intrinsic keeps it visible to the scheduler and the hazard recogniser)
extern "C" __device__ int tfx_writelane_i32(int val, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

// per-road descriptor words handed from phase M to phase C: ring indices / car counts
__device__ __forceinline__ int pack_idx(int ld, int lc) { return ld | (lc << 16); }
__device__ __forceinline__ int pack_cnt(int n_old, int n_tot) { return n_old | (n_tot << 16); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ unsigned long long lane_mask_below(int n) {  // lanes 0 .. n-1
  return n >= 64 ? ~0ull : (n <= 0 ? 0ull : ((1ull << n) - 1ull));
}

template <int CC, int S, int NBUF, int UNR, bool LEADER_LDS, int LIVE, int NP = 1>
__global__ __launch_bounds__(256) void k_move_dma(const Dev d, const int tidx) {
  static_assert(NBUF == 1 || CC > 0, "multi-buffering needs a compile-time capacity");
  static_assert(NBUF >= 1 && NBUF <= 3, "1 to 3 LDS buffers");
  static_assert(NP == 1 || (CC > 0 && !LEADER_LDS), "multi-pass roads: compile-time capacity, DPP leader");
  static_assert(CC == 0 || CC - 2 <= 64 * NP, "NP passes of 64 lanes must cover the ring");
  static_assert(LIVE == 0 || CC > 0, "live-chunk streaming needs a compile-time capacity");
  constexpr int TR = 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  // wave index made provably uniform: everything derived from it (tile, base pointers, road
  // counts) then lives in SGPRs and every branch on it is a scalar branch
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tick = *d.tickA;
  const int C = CC ? CC : d.C;
  const int sub_slots = S * C;  // (x, v) pairs per sub-tile
  float2 *bufs = reinterpret_cast<float2 *>(smem) + (size_t)wv * NBUF * sub_slots;
  constexpr int K_DMA = CC ? (S * CC * 8 + 1023) / 1024 : 0;  // DMA instructions per full sub-tile
  const bool all_slots = (C - 1 >= 64 * NP);  // every (pass, lane) ring position is a valid slot of the image

  // tiles are dealt round-robin over all waves of the grid: at any moment the waves in flight
  // cover one dense window of memory (DRAM rows are used while they are open)
  const long total = (long)d.E * d.R;
  const long tiles = (total + TR - 1) / TR;
  const long nw = (long)gridDim.x * 4;
  const int tick_sp = (d.spawn_mode == TFX_SPAWN_PERIODIC) ? tick % d.spawn_period : 0;
  const float ll = (lane == 0) ? 0.0f : d.car_l;  // the fake leader has length 0

  unsigned long long my_updates = 0;

  for (long tile = (long)blockIdx.x * 4 + wv; tile < tiles; tile += nw) {
    const long base = tile * TR;
    // ================= phase M: lane j <-> road base + j ========================================
    const bool valid = base + lane < total;
    const int id = valid ? (int)(base + lane) : (int)(total - 1);
    const int env = id / d.R;
    const int e = id - env * d.R;
    const bool frozen = valid && env_frozen(d, env, tick);
    const bool run = valid && !frozen;
    const RoadPrep p = prep_road(d, id, env, e, tick, tick_sp, tidx, run, run);
    const int pk = pack_idx(p.ld, p.lc), pn = pack_cnt(p.n_old, p.n_tot);
    // roads of this tile that received cars this tick (bit j = road j), wave-uniform
    const unsigned long long spawn_mask = __builtin_amdgcn_ballot_w64(run && p.n_tot != p.n_old);
    const unsigned long long frozen_mask = __builtin_amdgcn_ballot_w64(frozen);
    const int xL_bits = __float_as_int(p.xL), xs0_bits = __float_as_int(p.xs0);

    // ================= phase C: sub-tiles of S roads through LDS =================================
    const long left = total - base;
    const int nroads = left < TR ? (int)left : TR;
    float2 *tx = d.xv + (size_t)base * C;  // wave-uniform tile base, 16-byte aligned
    // road j's ballots, collected by lane j
    // (one set per pass of 64 cars for rings longer than a wavefront)
    int mw0[NP], mw1[NP], md0[NP], md1[NP], mp0[NP], mp1[NP], mf0[NP], mf1[NP], r_t = 0;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) mw0[pp] = mw1[pp] = md0[pp] = md1[pp] = mp0[pp] = mp1[pp] = mf0[pp] = mf1[pp] = 0;

    // LIVE > 0: only the 16-byte chunks (2 ring slots) that hold a live car are moved.  Bit i of
    // the returned mask says whether this lane's chunk of DMA instruction i is live; the same mask
    // drives the write-back (LIVE == 2).  Chunk q of a sub-tile belongs to road q / (C/2).
    auto issue = [&](int j0, float2 *dst) -> unsigned {
      const int ns = (nroads - j0 < S) ? nroads - j0 : S;
      const float2 *gsrc = tx + (size_t)j0 * C;
      const int n16 = (ns * C) >> 1;  // 16-byte chunks
      unsigned bits = 0;
      int i = 0;
      for (int q0 = 0; q0 < n16; q0 += 64, ++i) {
        const int q = q0 + lane;
        bool go = q < n16;
        if (LIVE) {
          const int half = C >> 1;
          const int jj = q / half;
          const int s0 = 2 * (q - jj * half);                 // slots s0, s0 + 1
          const int ldr = __shfl(pk, (j0 + jj) & 63, 64) & 0xffff;
          const int nr = (int)((unsigned)__shfl(pn, (j0 + jj) & 63, 64) >> 16);
          int p0 = s0 - ldr - 1;                              // ring position of slot s0 behind the leader
          p0 += (p0 < 0) ? C - 1 : 0;
          int p1 = s0 - ldr;
          p1 += (p1 < 0) ? C - 1 : 0;
          go = go && ((s0 > 0 && p0 < nr) || p1 < nr);
          bits |= go ? (1u << i) : 0u;
        }
        if (go)
          __builtin_amdgcn_global_load_lds((gptr_t *)(gsrc + (size_t)q * 2), (lptr_t *)(dst + (size_t)q0 * 2), 16, 0, 0);
      }
      return bits;
    };

    // Prefetch distance D = NBUF - 1 sub-tiles.  live[k] = live-chunk mask of the sub-tile that
    // sits in buffer k.
    constexpr int D = NBUF - 1;
    unsigned live[NBUF];
#pragma unroll
    for (int k = 0; k < NBUF; ++k) live[k] = 0;
    const int nsub = (nroads + S - 1) / S;
    const int nfull = nroads / S;  // sub-tiles 0 .. nfull-1 are full
    if (D > 0) {
#pragma unroll
      for (int k = 0; k < D; ++k)
        if (k < nsub) live[k] = issue(k * S, bufs + k * sub_slots);
    }
    int slotb = 0;  // buffer of the current sub-tile = sub % NBUF
    for (int sub = 0; sub < nsub; ++sub) {
      const int j0 = sub * S;
      const int ns = (nroads - j0 < S) ? nroads - j0 : S;
      float2 *buf = bufs + slotb * sub_slots;
      unsigned live_cur;
      if (D > 0) {
        const int nx = sub + D;                     // sub-tile to put in flight now
        int nb = slotb + D;
        nb -= (nb >= NBUF) ? NBUF : 0;
        if (nx < nsub) {
          const unsigned lv = issue(nx * S, bufs + nb * sub_slots);
#pragma unroll
          for (int k = 0; k < NBUF; ++k)
            if (k == nb) live[k] = lv;
        }
        // younger than the loads of `sub`: the D sub-tiles issued after it and, once the pipeline
        // is full, the D write-backs in between - all of K_DMA instructions when the sub-tiles
        // involved are full; otherwise drain
        if (nx < nfull && sub >= D) wait_vmcnt<2 * D * K_DMA>();
        else if (nx < nfull) wait_vmcnt<D * K_DMA>();
        else wait_vmcnt<0>();
        live_cur = 0;
#pragma unroll
        for (int k = 0; k < NBUF; ++k)
          if (k == slotb) live_cur = live[k];
      } else {
        live_cur = issue(j0, buf);
        wait_vmcnt<0>();
      }
      slotb = (slotb + 1 >= NBUF) ? 0 : slotb + 1;

      // One road: lane k <-> the k-th car behind the fake leader.  SPAWN = false is the common case
      // (no car entered this road this tick): straight-line code, so the unrolled bodies of a
      // sub-tile form one basic block and the compiler interleaves independent roads.
      auto road = [&](int jj, auto spawn_tag) {
        constexpr bool SPAWN = decltype(spawn_tag)::value;
        const int j = j0 + jj;
        const int pkj = __builtin_amdgcn_readlane(pk, j), pnj = __builtin_amdgcn_readlane(pn, j);
        const float xLj = __int_as_float(__builtin_amdgcn_readlane(xL_bits, j));
        const int ldj = pkj & 0xffff, lcj = (int)((unsigned)pkj >> 16);
        const int n_oldj = pnj & 0xffff, n_totj = (int)((unsigned)pnj >> 16);
        float2 *rb = buf + jj * C;  // this road's record in LDS
        const unsigned lc_seg2 = (ldj > lcj) ? (unsigned)lcj : 0u;
        // leader of the first car of a pass: the fake leader, then the last car of the pass before
        float lead_x = xLj, lead_v = 0.0f;
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
          const int kk = pp * 64 + lane;  // car index behind the fake leader
          if (NP > 1 && pp > 0 && pp * 64 >= n_totj) break;  // wave-uniform: no cars in this pass
          // ring slot of car kk: positions run 0 .. C-2 from slot 1
          const unsigned pos = (unsigned)(ldj + kk);  // (ld - 1) + (kk + 1)
          const unsigned slot = 1u + min(pos, pos - (unsigned)(C - 1));
          const bool in_img = all_slots || kk < C - 1;
          const unsigned sl = in_img ? slot : 1u;
          const float2 cv = rb[sl];
          float x = cv.x, v = cv.y;
          if (SPAWN) {  // cars spawned this tick are not in memory yet
            if (kk >= n_oldj && kk < n_totj) {
              x = spawned_x(d, __int_as_float(__builtin_amdgcn_readlane(xs0_bits, j)), kk - n_oldj);
              v = d.car_v;
            }
          }
          float xl, vl;
          if (LEADER_LDS && !SPAWN) {
            // the car one ring slot ahead in the image; lane 0's leader is the fake one
            const unsigned prev = (sl == 1u) ? (unsigned)(C - 1) : sl - 1u;
            const float2 lv = rb[prev];
            xl = (lane == 0) ? xLj : lv.x;
            vl = (lane == 0) ? 0.0f : lv.y;
          } else {
            // wave_shr:1 - lane k receives lane k-1; lane 0 keeps `old` = the leader of this pass
            xl = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lead_x), __float_as_int(x), 0x138, 0xf, 0xf, false));
            vl = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lead_v), __float_as_int(v), 0x138, 0xf, 0xf, false));
          }
          if (NP > 1) {  // OLD state of this pass's last car leads the next pass (Jacobi)
            lead_x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
            lead_v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
          }
          const float llp = (pp == 0) ? ll : d.car_l;  // only the fake leader has length 0

          float xn, vn;
          // the reciprocal form of the two constant-divisor divisions is exact on a verified
          // domain; one live car outside it (denormal-range or non-finite speed) sends the pass
          // down the IEEE-divide path.  Wave-uniform branch.
          const unsigned long long off_domain =
              __builtin_amdgcn_ballot_w64(!idm_fast_domain(v)) & lane_mask_below(n_totj - pp * 64);
          if (d.fastdiv && off_domain == 0ull) {
            idm_step_fast(d, x, v, xl, vl, llp, xn, vn);
          } else {
            idm_step(d, x, v, xl, vl, llp, xn, vn);
          }

          // new state back into the image: lanes without a car rewrite dead slots with junk,
          // which nothing reads (the lanes cover every ring slot except `leading`)
          if (in_img) rb[sl] = make_float2(xn, vn);
          if (SPAWN && d.w) {
            if (kk >= n_oldj && kk < n_totj) d.w[(size_t)base * C + j * C + slot] = (float)tick;
          }
          // wrapped ring: the reference tests x, not v, on the second segment (:210)
          const float wq = (slot <= lc_seg2) ? xn : vn;
          const unsigned long long m_wait = __builtin_amdgcn_ballot_w64(wq < d.thresh);
          const unsigned long long m_det = __builtin_amdgcn_ballot_w64(xn > d.near_end);
          const unsigned long long m_pop = __builtin_amdgcn_ballot_w64(xn > d.length);
          const unsigned long long m_far = __builtin_amdgcn_ballot_w64((xn - d.length) > d.length);
          mw0[pp] = tfx_writelane_i32((int)(unsigned)m_wait, j, mw0[pp]);
          mw1[pp] = tfx_writelane_i32((int)(unsigned)(m_wait >> 32), j, mw1[pp]);
          md0[pp] = tfx_writelane_i32((int)(unsigned)m_det, j, md0[pp]);
          md1[pp] = tfx_writelane_i32((int)(unsigned)(m_det >> 32), j, md1[pp]);
          mp0[pp] = tfx_writelane_i32((int)(unsigned)m_pop, j, mp0[pp]);
          mp1[pp] = tfx_writelane_i32((int)(unsigned)(m_pop >> 32), j, mp1[pp]);
          mf0[pp] = tfx_writelane_i32((int)(unsigned)m_far, j, mf0[pp]);
          mf1[pp] = tfx_writelane_i32((int)(unsigned)(m_far >> 32), j, mf1[pp]);
          // x of the last car after the move (junk, and unused, when the road is empty)
          const int tl = n_totj - 1 - pp * 64;
          if (NP == 1 || (tl >= 0 && tl < 64))
            r_t = tfx_writelane_i32(__builtin_amdgcn_readlane(__float_as_int(xn), tl > 0 ? tl : 0), j, r_t);
        }
      };

      const unsigned long long sub_spawn = (spawn_mask >> j0) & ((1ull << S) - 1ull);
      const unsigned long long sub_frozen = (frozen_mask >> j0) & ((1ull << S) - 1ull);
      if (ns == S && (sub_spawn | sub_frozen) == 0ull) {
        for (int jj0 = 0; jj0 < S; jj0 += UNR) {
#pragma unroll
          for (int ju = 0; ju < UNR; ++ju) road(jj0 + ju, std::false_type{});
        }
      } else {
        for (int jj = 0; jj < ns; ++jj) {
          if ((sub_frozen >> jj) & 1ull) continue;  // the road's image goes back unchanged
          if ((sub_spawn >> jj) & 1ull) road(jj, std::true_type{});
          else road(jj, std::false_type{});
        }
      }
      {
        // write the sub-tile back as it came: 16 B per lane, 1 KiB per wave instruction
        __builtin_amdgcn_wave_barrier();
        const float4 *src4 = reinterpret_cast<const float4 *>(buf);
        float4 *dst4 = reinterpret_cast<float4 *>(tx + (size_t)j0 * C);
        const int n16 = (ns * C) >> 1;
        int i = 0;
        for (int q0 = 0; q0 < n16; q0 += 64, ++i) {
          const int q = q0 + lane;
          if (q < n16 && (LIVE < 2 || ((live_cur >> i) & 1u))) dst4[q] = src4[q];
        }
      }
    }

    // ================= phase W: lane j finishes road j ==========================================
    if (run) {
      // cars popped from the head: the while loop (:123) stops at the first car still on the road
      int kpop = 0, n_wait = 0, n_det = 0;
      bool open = true, any_far = false;
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        const unsigned long long live = lane_mask_below(p.n_tot - pp * 64);
        const unsigned long long m_pop = ((unsigned long long)(unsigned)mp1[pp] << 32 | (unsigned)mp0[pp]) & live;
        const unsigned long long m_wait = ((unsigned long long)(unsigned)mw1[pp] << 32 | (unsigned)mw0[pp]) & live;
        const unsigned long long m_det = ((unsigned long long)(unsigned)md1[pp] << 32 | (unsigned)md0[pp]) & live;
        const unsigned long long m_far = ((unsigned long long)(unsigned)mf1[pp] << 32 | (unsigned)mf0[pp]) & m_pop;
        const int lead = (~m_pop == 0ull) ? 64 : __builtin_ctzll(~m_pop);
        if (open) kpop += lead;
        open = open && lead == 64;  // the whole pass left: the prefix continues in the next one
        n_wait += __popcll(m_wait);
        n_det += __popcll(m_det);
        any_far = any_far || (m_far != 0ull);
      }
      if (e < d.r) {
        int *ob = d.obs + (size_t)env * d.obs_len;
        if (p.n_tot > 0) {
          d.waiting[(size_t)env * d.r + e] += n_wait;
          ob[d.r + e] = n_det;
        }
        ob[e] = (d.agent_mode && tidx > 0) ? ob[e] + kpop : kpop;
        if (kpop > 0) d.passed_dst[(size_t)env * d.I + e % d.I] = 1;
      }
      d.rec[id] = make_int4(rec_pack(kpop, p.ld, C), p.ovf_sp, r_t, p.n_tot);
      if (needs_serial(kpop, any_far, p.n_tot, C)) d.env_flag[env] = tick + 1;
      // the leader's x stays in its slot (after the write-back of the image, same wave)
      d.xv[(size_t)id * C + p.ld].x = p.xL;
      d.leadx[id] = p.xL;  // k_advance re-installs it in the slot a pop frees (advance_road)
      my_updates += (unsigned long long)p.n_tot;
    }
  }

  for (int off = 32; off > 0; off >>= 1) my_updates += __shfl_down(my_updates, off);
  if (lane == 0 && my_updates) veh_add(d.veh, my_updates);
  if (blockIdx.x == 0 && threadIdx.x == 0) *d.tickB = tick;
}

}  // namespace tfx
