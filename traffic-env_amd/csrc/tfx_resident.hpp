// tfx_resident.hpp - k_res: MANY ticks of the env step in ONE launch for envs small enough to live
// in a compute unit's LDS (cfg0, cfg1, the reference's own 3x3 default: R * (C-1) * 8 bytes per env).
//
// The per-tick kernels stream every car through HBM twice per tick and pay two launches per tick;
// for small grids that is all latency (round 1: cfg1 x 1024 envs at 0.18 of the HBM roofline, 38 us
// per tick of which the cars need 3).  Here a workgroup owns `epb` whole envs for the whole call:
//   * LPR = 1, 2 or 4 lanes per road (lanes = LPR * epb * R, packed across env boundaries so wavefronts
//     stay full), or the mixed form (3): two lanes per road, four on the roads cars enter the map on.  With two lanes the road's cars split in halves - the Jacobi update needs only OLD
//     neighbours, so the second lane starts from the OLD state of the car in front of its half (read
//     before the first lane, which sits next to it in the same wavefront, overwrites it) - and the
//     walk, the tick's critical path, is half as long; pops, counts and the tail meet through a lane
//     shuffle (the second half's pop prefix counts only if every car of the first half popped);
//   * the cars sit in LDS in the REFERENCE's ring layout - ring[slot][lane], slots 1..C-1, the same
//     leading / lastcar indices as traffic_env.py - so a tick moves no car that did not move: the IDM
//     walk (k_move_t's lane-per-road chain: the leader of car k is the lane's previous car, OLD
//     values, Jacobi) rewrites slots in place, pops advance `leading`, pushes advance `lastcar`;
//   * a tick is: lights + spawns + walk | barrier | handoff | barrier.  The handoff is the pull form
//     of tfx_advance.hpp (each road takes the cars its unique predecessor popped, under the
//     reference's road-order rule) in two phases - every lane first copies its predecessor's popped
//     cars to registers, barrier, then pushes them onto its own ring - which makes it exact for any
//     number of pops and any ring fill (a push can only overwrite a popped car that was copied in
//     the same or an earlier round).  Only a car that leaves TWO roads in one tick sends its env
//     through the literal serial loop (one lane, on the LDS image);
//   * the agent step's `if done: break` (traffic_test.py:55) is a per-env flag in LDS: the lanes of
//     an env that overflowed idle through the remaining ticks of the call;
//   * HBM is touched at the start (cars, indices, light words) and at the end (the same, plus obs,
//     rewards, waiting, passed_dst); per-tick inputs (actions, arrival counts) are a few words.
// Arithmetic, ring indices, counters: bit-identical to the per-tick kernels and the oracle
// (tests/test_gpu_fused.py runs every case through both).
#pragma once
#include "tfx_common.hpp"
#include "tfx_move_t.hpp"
#include "tfx_idm_pair.hpp"
#include "tfx_misc.hpp"

namespace tfx {

struct ResArgs {
  int epb;             // envs per workgroup
  int n_ticks;
  int greedy_spacing;  // > 0: the greedy controller decides in the kernel (algorithms/greedy.py:14-16)
  int *greedy_act;     // [E][I] held greedy actions (in/out)
  int own_clock;       // single-workgroup launch: the kernel advances the device clock itself
  // the rest of an agent decision (tfx_agent_step), folded into the kernel's tail when `tail` is set:
  // remi_reward() (traffic_env.py:64-78) if `remi`, the Repeater's observation (traffic_test.py:48-53),
  // the decision's rewards and done flags - what k_remi, k_agent_obs, a copy and k_done_since do after
  // the per-tick kernels
  int tail, remi;
  float *aobs, *areward;  // [E][2r+I], [E][I] (may be null)
  uint8_t *adone;         // [E] (may be null)
  // on-device Poisson arrivals (tfx_set_poisson): the stream of k_poisson, drawn inside the kernel
  int poisson;
  PoissonDev ps;
  // LPR = 3 (two lanes per road, four on the roads without a predecessor): columns of the workgroup's LDS arrays, and
  // how many of an env's road slots are interior roads / roads without a predecessor (build_slots' order)
  int cols, n_int, n_ent;
};

constexpr int RES_KH = 2;  // popped cars copied per handoff round
constexpr int RES_MAX_THREADS = 512;  // lanes (= roads) per workgroup: 8 wavefronts, up to 256 VGPRs each

// bytes of dynamic LDS for Tr road columns (Tr = lanes / lanes-per-road)
__host__ __device__ inline size_t res_lds_bytes(int Tr, int C, int epb, int I, int n_entry, bool W) {
  const size_t ns = (size_t)(C - 1);
  size_t b = ns * Tr * sizeof(float2);
  if (W) b += ns * Tr * sizeof(float);
  b += (size_t)Tr * 8 * 4;               // ld, lc, kpop, cnt, tail, passed, ovf, ovfsp
  b += (size_t)epb * I * (2 * 8 + 4 + 4 + 4);  // light[2], rew, pdst, act
  b += (size_t)epb * 2 * 4 + 16;         // ovftick, fartick, maxpop[2]
  b += (size_t)epb * (n_entry + 2) * 4;  // Poisson: this tick's cars per entry road, the stream's state
  return (b + 15) & ~(size_t)15;
}

// At most 128 VGPRs (4 wavefronts per SIMD; the compiler spills ~10 registers to do it): with the 140 it
// would take the kernel fits exactly 3 wavefronts per SIMD, i.e. exactly the 12 wavefronts of the four
// 3-wavefront workgroups a CU gets at cfg1 x 1024 - but only if the dispatcher spreads them 3-3-3-3 over the
// SIMDs.  After idleness or a different kernel it did not: a CU then held three workgroups, the fourth ran
// in a second round and the whole launch took 1.45x (rocprofv3: same clock, same wave-cycles; 12 launches of
// 36 in a loop that interleaves another kernel, 0 of 36 with this bound, and the steady state is no slower).
// Round 4 re-measured (make exp EXP=RW3:-DRES_WAVES=3 ...), cfg1 x 1024, a 10-tick call at 3 / 4 / 5 / 6 wavefronts per SIMD:
// 100 / 103 / 117 / 133 us - and at 3 the bench's regions spread over 8.5 .. 13.7 us per tick (the misplaced launches)
// against 8.5 .. 9.6 at 4; four lanes per road at 5 per SIMD (all four workgroups of a CU side by side): 150 us.
template <int LPR, bool W>
#ifndef RES_WAVES
#define RES_WAVES 4
#endif
__global__ __launch_bounds__(RES_MAX_THREADS) __attribute__((amdgpu_waves_per_eu(RES_WAVES, 8)))
void k_res(const Dev d, const ResArgs a) {
  static_assert(LPR >= 1 && LPR <= 4, "one, two or four lanes per road (adjacent lanes of one wavefront); 3: two, and four on entry roads");
  // LPR = 3, the mixed form: the roads a tick waits for are the longest ones, and under load those are the roads cars
  // enter the map on (their queues reach the ring's capacity while an interior road holds what a green phase lets in).
  // They get four lanes, every other road two: at cfg1 the longest chain of a tick drops from 16 cars to 12 with the same
  // three wavefronts per env (160 + 32 lanes) - the whole-env form of four lanes per road needs five.
  constexpr bool MIX = LPR == 3;
  constexpr int LMAX = MIX ? 4 : LPR;  // most lanes any road has
  extern __shared__ __align__(16) unsigned char res_smem[];
  const int C = d.C, NS = C - 1, R = d.R, I = d.I;
  const int T = MIX ? a.cols : (int)blockDim.x / LMAX;  // T: road columns of the workgroup
  const int epb = a.epb;
  float2 *ring = reinterpret_cast<float2 *>(res_smem);
  float *ringw = reinterpret_cast<float *>(ring + (size_t)NS * T);
  int *s_ld = reinterpret_cast<int *>(W ? (ringw + (size_t)NS * T) : ringw);
  int *s_lc = s_ld + T, *s_kpop = s_lc + T, *s_cnt = s_kpop + T;
  float *s_tail = reinterpret_cast<float *>(s_cnt + T);
  int *s_passed = reinterpret_cast<int *>(s_tail + T), *s_ovf = s_passed + T, *s_ovfsp = s_ovf + T;
  int2 *s_light = reinterpret_cast<int2 *>(s_ovfsp + T);  // [2][epb * I]
  float *s_rew = reinterpret_cast<float *>(s_light + 2 * (size_t)epb * I);
  int *s_pdst = reinterpret_cast<int *>(s_rew + (size_t)epb * I);
  int *s_act = s_pdst + (size_t)epb * I;
  int *s_ovftick = s_act + (size_t)epb * I;  // [epb] tick + 1 of the env's last overflow in this call
  int *s_fartick = s_ovftick + epb;          // [epb] tick + 1 of the last tick that needs the serial loop
  int *s_maxpop = s_fartick + epb;           // [2] most cars any road popped, by tick parity
  int *s_spawn = s_maxpop + 4;               // [epb][n_entry] Poisson arrivals of the tick
  int *s_gap = s_spawn + (size_t)epb * d.n_entry;  // [epb] whole ticks until the env's next car (-1: not drawn yet)
  unsigned *s_draws = reinterpret_cast<unsigned *>(s_gap + epb);  // [epb] index of the env's next car

  int lpr = LMAX, lsh = LMAX == 4 ? 2 : (LMAX == 2 ? 1 : 0);  // this road's lanes, and their log2
  int h = (int)(threadIdx.x & (LMAX - 1));  // which share (half, quarter) of the road's cars this lane walks
  int t = (int)(threadIdx.x / LMAX);        // the road's column
  if (MIX) {
    // an env's lanes, in the slot order of its roads: 2 per interior road | 4 per road without a predecessor | 2 per exit
    // road (2 n_int and the lanes of an env are multiples of 4: a road's lanes are adjacent lanes of one wavefront)
    const int per_env = 2 * R + 2 * a.n_ent;
    const int el0 = (int)threadIdx.x / per_env, q = (int)threadIdx.x - el0 * per_env;
    const int a0 = 2 * a.n_int, a1 = a0 + 4 * a.n_ent;
    int col;
    lpr = 2;
    lsh = 1;
    if (q < a0) {
      col = q >> 1;
      h = q & 1;
    } else if (q < a1) {
      col = a.n_int + ((q - a0) >> 2);
      h = (q - a0) & 3;
      lpr = 4;
      lsh = 2;
    } else {
      col = a.n_int + a.n_ent + ((q - a1) >> 1);
      h = (q - a1) & 1;
    }
    t = el0 * R + col;
    if (t >= T) t = T - 1;  // (lanes past the last env: a column of their own, never a valid road)
  }
  const bool hA = h == 0;                                // the road's first lane also does everything per road
  // Columns follow the storage-slot order of the env's roads (interior train roads, entry roads, exit
  // roads: build_slots) rather than road ids: roads of a kind have similar car counts, a wavefront
  // walks as far as its longest road, and a wavefront of short roads frees its SIMD early for the
  // other workgroups of the CU.
  const int env_l = t / R, rc = t - env_l * R;  // the road's column within its env
  const int cbase = t - rc;                      // column of the env's first road slot
  const int env = blockIdx.x * epb + env_l;
  const int e = (env_l < epb && env < d.E) ? d.slot_road[rc] : 0;
  auto COL = [&](int road) { return cbase + d.road_slot[road]; };
  const bool valid = env_l < epb && env < d.E;
  const int id = valid ? env * R + e : 0;
  const int tick0 = *d.tickA;  // advanced by k_tick_add after this kernel (inside it only when it is one workgroup)
  const bool train = valid && e < d.r;
  const int dir = train ? e / I : 0;
  const int isec = train ? e - dir * I : 0;
  const int li = env_l * I + isec;  // index of the dest intersection in the per-block arrays
  const int phase_e = (dir < 2) ? 1 : 0;  // roadgraph.py:36
  const int ej = valid ? d.entry_idx[e] : -1;
  const int pe = valid ? d.pred[e] : -1;
  const int nx = train ? d.nexts[e] : -1;
  const int tn = nx >= 0 ? COL(nx) : 0, tp = pe >= 0 ? COL(pe) : 0;  // columns of the next / previous road
  // columns of the four roads into intersection `isec` (greedy rule, rewards, remi)
  const int c4[4] = {train ? COL(isec) : 0, train ? COL(I + isec) : 0, train ? COL(2 * I + isec) : 0, train ? COL(3 * I + isec) : 0};
  const bool des = train && dir == 0 && hA;    // this lane also keeps intersection `isec`
  auto RG = [&](int slot, int lane) -> float2 & { return ring[(size_t)(slot - 1) * T + lane]; };
  auto RW = [&](int slot, int lane) -> float & { return ringw[(size_t)(slot - 1) * T + lane]; };

  // ---- load: ring indices, cars (either global layout -> ring slots), light words ---------------
  int ld = 1, lc = 1;
  if (valid) {
    ld = d.leading[id];
    lc = d.lastcar[id];
    const int n = ring_count(ld, lc, C);
    const int head = wrap1(ld + 1, C);
    if (d.layout == 1) {
      const size_t col = tcol(d, env, e);
      for (int k = h; k < n; k += lpr) {  // (the road's lanes share the copy)
        const int slot = ring_adv(head, k, C);
        RG(slot, t) = d.xv[col + (size_t)k * 64];
        if (W) RW(slot, t) = d.w[col + (size_t)k * 64];
      }
    } else {
      for (int k = h; k < n; k += lpr) {
        const int slot = ring_adv(head, k, C);
        RG(slot, t) = d.xv[(size_t)id * C + slot];
        if (W) RW(slot, t) = d.w[(size_t)id * C + slot];
      }
    }
  }
  if (valid && hA) {
    const int n = ring_count(ld, lc, C);
    s_ld[t] = ld;
    s_lc[t] = lc;
    s_cnt[t] = n;
    s_tail[t] = d.tailx[id];
    s_kpop[t] = 0;
    s_ovf[t] = 0;
    if (train) s_passed[t] = d.obs[(size_t)env * d.obs_len + e];
    if (des) {
      const int *ob = d.obs + (size_t)env * d.obs_len + 2 * d.r;
      s_light[(size_t)epb * I + li] = make_int2(ob[isec], ob[I + isec]);  // parity 1: read by tick 0
      s_rew[li] = d.rewards[(size_t)env * I + isec];
      s_pdst[li] = 0;
      s_act[li] = a.greedy_spacing > 0 ? a.greedy_act[(size_t)env * I + isec] : 0;
    }
    if (e == 0) {
      s_ovftick[env_l] = 0;
      s_fartick[env_l] = 0;
    }
  }
  if (threadIdx.x == 0) s_maxpop[0] = s_maxpop[1] = 0;
  if (a.poisson && (int)threadIdx.x < epb && (int)(blockIdx.x * epb + threadIdx.x) < d.E) {
    s_gap[threadIdx.x] = a.ps.gap_left[blockIdx.x * epb + threadIdx.x];
    s_draws[threadIdx.x] = a.ps.draws[blockIdx.x * epb + threadIdx.x];
  }
  __syncthreads();

  // The on-device rules divide by run-time periods: ((tick + env % P) / P) & 1, tick % P, e % P.  The
  // quotients and remainders are carried from tick to tick instead (a division is ~40 instructions).
  const int cyc_P = d.action_period > 0 ? d.action_period : 1;
  int cyc_rem = 0, cyc_q = 0;
  if (d.action_mode == TFX_ACTION_CYCLE && a.greedy_spacing <= 0) {
    const int s0 = tick0 + (env + d.env_off) % cyc_P;
    cyc_q = s0 / cyc_P;
    cyc_rem = s0 - cyc_q * cyc_P;
  }
  const int sp_P = d.spawn_period > 0 ? d.spawn_period : 1;
  const int sp_phase = e % sp_P;  // entry road e gets a car when tick % P == e % P
  int sp_rem = tick0 % sp_P;
  const int gr_P = a.greedy_spacing > 0 ? a.greedy_spacing : 1;
  int gr_rem = tick0 % gr_P;
  int wait_acc = 0, det = 0;
  bool det_set = false;
  float xL = INFINITY;
  bool xL_set = false;
  unsigned long long my_updates = 0;

  for (int tt = 0; tt < a.n_ticks; ++tt) {
    const int tick = tick0 + tt, par = tt & 1;
    const bool gr_now = gr_rem == 0;  // a tick in which the greedy controller decides
    // env stopped for the rest of the agent step (`if done: break`): it overflowed earlier in this call
    const bool frozen = d.agent_mode && valid && s_ovftick[env_l] > tick0;
    const bool run = valid && !frozen;
    int kpop = 0, n_tot = 0, ovf_sp = 0;
    float tail_x = 0.0f;
    bool far = false;
    if (a.poisson) {
      // The reference's Poisson generator with the device RNG, exactly as k_poisson (tfx_misc.hpp) draws
      // it: one wavefront per env evaluates 64 consecutive cars at a time; every car up to and
      // including the first with a non-zero gap arrives in this tick.
      const int lane = threadIdx.x & 63, n_waves = blockDim.x >> 6;
      for (int el = threadIdx.x >> 6; el < epb; el += n_waves) {
        const int envp = blockIdx.x * epb + el;
        int *hist = s_spawn + (size_t)el * d.n_entry;
        for (int j = lane; j < d.n_entry; j += 64) hist[j] = 0;
        __builtin_amdgcn_wave_barrier();
        if (envp < d.E && !(d.agent_mode && s_ovftick[el] > tick0)) {
          int gap = s_gap[el];
          unsigned c0 = s_draws[el];
          const unsigned gid = (unsigned)(envp + d.env_off);
          unsigned u[4];
          auto gap_of = [&](unsigned draw) {
            philox4x32(draw, gid, 0x47415021u, 0u, a.ps.seed_lo, a.ps.seed_hi, u);
            int k = 0;
            while (k < a.ps.n_cdf - 1 && u[0] >= a.ps.cdf[k]) ++k;
            return k;
          };
          if (a.ps.regular) {  // the `regular` generator, as k_poisson runs it
            const bool due = a.ps.every == 0 || gap % a.ps.every == 0;
            ++gap;
            if (due) {
              for (int j = lane; j < a.ps.burst; j += 64) {
                philox4x32(1u + 2u * (c0 + (unsigned)j), gid, 0x524F4144u, 0u, a.ps.seed_lo, a.ps.seed_hi, u);
                atomicAdd(&hist[(int)(((unsigned long long)u[0] * (unsigned)d.n_entry) >> 32)], 1);
              }
              c0 += (unsigned)a.ps.burst;
            }
          } else {
            if (gap < 0) gap = gap_of(0u);
            if (gap > 0) {
              --gap;
            } else {
              for (int guard = 0; guard < 1024; ++guard) {
                const unsigned c = c0 + (unsigned)lane;
                const int g = gap_of(2u + 2u * c);
                const unsigned long long stop = __builtin_amdgcn_ballot_w64(g > 0);
                const int f = stop ? __builtin_ctzll(stop) : 63;
                if (lane <= f) {
                  philox4x32(1u + 2u * c, gid, 0x524F4144u, 0u, a.ps.seed_lo, a.ps.seed_hi, u);
                  atomicAdd(&hist[(int)(((unsigned long long)u[0] * (unsigned)d.n_entry) >> 32)], 1);
                }
                c0 += (unsigned)(f + 1);
                if (stop) {
                  gap = __shfl(g, f, 64) - 1;
                  break;
                }
              }
            }
          }
          if (lane == 0) {
            s_gap[el] = gap;
            s_draws[el] = c0;
          }
        }
      }
      __syncthreads();
    }
    if (des && frozen) {
      s_light[(size_t)par * epb * I + li] = s_light[(size_t)(par ^ 1) * epb * I + li];
      // (the per-tick path's k_greedy keeps deciding for a stopped env, from its standing counts)
      if (a.greedy_spacing > 0 && gr_now) {
        s_act[li] = (s_cnt[c4[0]] + s_cnt[c4[1]] - s_cnt[c4[2]] - s_cnt[c4[3]] < 0) ? 1 : 0;
      }
    }
    if (run && hA) {
      ld = s_ld[t];
      lc = s_lc[t];
      const int n_old = ring_count(ld, lc, C);
      n_tot = n_old;
      // ---- lights (TrafficEnv._step :225-232, update_lights :81-94) ------------------------------
      xL = INFINITY;
      xL_set = true;
      if (train) {
        const int2 pl = s_light[(size_t)(par ^ 1) * epb * I + li];
        int ph_new, el_new;
        if (a.greedy_spacing > 0) {
          // greedy: every `spacing` ticks phase 1 iff the two N-S approaches hold more cars than the
          // two E-W ones (cars_on_roads().dot([1,1,-1,-1]) < 0); held in between
          int act = s_act[li];
          if (gr_now) {
            act = (s_cnt[c4[0]] + s_cnt[c4[1]] - s_cnt[c4[2]] - s_cnt[c4[3]] < 0) ? 1 : 0;
          }
          int change;
          if (d.learn_switch) { change = act != 0; ph_new = ((pl.x != 0) != (act != 0)) ? 1 : 0; }
          else { change = (pl.x != 0) != (act != 0); ph_new = act; }
          el_new = change ? 0 : pl.y + 1;
          if (des && gr_now) s_act[li] = act;  // (read by the others in later ticks only)
        } else if (d.action_mode == TFX_ACTION_CYCLE) {  // light_next with the carried quotient
          const int act = cyc_q & 1;
          int change;
          if (d.learn_switch) { change = act != 0; ph_new = ((pl.x != 0) != (act != 0)) ? 1 : 0; }
          else { change = (pl.x != 0) != (act != 0); ph_new = act; }
          el_new = change ? 0 : pl.y + 1;
        } else {
          light_next(d, env, isec, tick, tt, pl.x, pl.y, ph_new, el_new);
        }
        if (des) s_light[(size_t)par * epb * I + li] = make_int2(ph_new, el_new);
        if (phase_e == ph_new || el_new < d.yellow) xL = d.length;
        else if (s_cnt[tn] > 0) xL = s_tail[tn] + d.length;
      }
      // ---- spawns (add_new_cars :274-283 -> add_car :97-114): straight into the ring -------------
      tail_x = s_tail[t];
      if (ej >= 0) {
        const int c = a.poisson ? s_spawn[(size_t)env_l * d.n_entry + ej]
                      : (d.spawn_mode == TFX_SPAWN_PERIODIC ? (sp_rem == sp_phase ? 1 : 0) : spawn_count(d, env, e, ej, 0, tt));
        for (int q = 0; q < c; ++q) {
          const int pos = wrap1(lc + 1, C);
          const float start = (lc != ld) ? (tail_x - d.car_l) - d.car_s0 : INFINITY;
          if (pos != ld) {
            const float xs = (start < 0.0f) ? start : 0.0f;  // min(car.x = 0, start)
            RG(pos, t) = make_float2(xs, d.car_v);
            if (W) RW(pos, t) = (float)tick;
            ++n_tot;
            lc = pos;
            tail_x = xs;
          } else {
            ++ovf_sp;
          }
        }
      }
    }
    // ---- move_cars (:187-212): the lane walks its road from the head, leader chain in registers ---
    {
      if (LMAX > 1) {  // what the road's first lane worked out, for the others
        const int src = (int)(threadIdx.x & 63u & ~(unsigned)(lpr - 1));
        n_tot = __shfl(n_tot, src, 64);
        ld = __shfl(ld, src, 64);
        lc = __shfl(lc, src, 64);
        xL = __shfl(xL, src, 64);
      }
      // this lane's share of the road: cars [my_k0, my_k0 + my_n)
      const int n_share = (n_tot + lpr - 1) >> lsh;
      const int my_k0 = h * n_share;
      const int my_n = run ? (n_tot - my_k0 < 0 ? 0 : (n_tot - my_k0 < n_share ? n_tot - my_k0 : n_share)) : 0;
      float xprev = xL, vprev = 0.0f, llv = 0.0f;
      const int head = wrap1(ld + 1, C);
      if (LMAX > 1 && !hA && my_n > 0) {
        // a later share follows the last car of the share before it: its OLD state, read here - before the
        // neighbouring lane (same wavefront, so in program order) rewrites that slot
        const float2 lead = RG(ring_adv(head, my_k0 - 1, C), t);
        xprev = lead.x;
        vprev = lead.y;
        llv = d.car_l;
      }
      int n_wait = 0, n_det = 0;
      bool open = true;  // (second half: provisionally - it counts only if the whole first half popped)
      // wrapped ring: x, not v, is tested on slots 1..lastcar (:210) = the cars from index kq on
      const int kq = (ld > lc) ? C - 1 - ld : 0x7fffffff;
      // Groups of G cars computed together, as in k_move_t2: the kernel runs one or two wavefronts
      // per SIMD, so what it waits for is the dependent-instruction chain of a car's IDM step -
      // cars k and k+1 are independent (both read OLD values), their arithmetic issues packed
      // (v_pk_*_f32 on the pair) and the pairs of a group interleave; one domain test per group
      // decides between idm_pair and the literal idm_step.  Lanes past their road's end compute on
      // zeros; only the ring writes and the counters are predicated.
      constexpr int G = 4;
      const bool fast_ok = d.fastdiv && d.fastmax;
      int slot = ring_adv(head, my_k0, C);
      int psl[G];
      float2 pf[G];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        psl[u] = slot;
        pf[u] = (u < my_n) ? RG(slot, t) : make_float2(0.0f, 0.0f);
        slot = wrap1(slot + 1, C);
      }
      for (int k0 = 0; __builtin_amdgcn_ballot_w64(k0 < my_n) != 0ull; k0 += G) {  // while any lane has cars left
        float2 cur[G];
        int csl[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
          cur[u] = pf[u];
          csl[u] = psl[u];
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {  // the next group's cars (per-lane guards: nothing is read past a lane's share)
          psl[u] = slot;
          pf[u] = (k0 + G + u < my_n) ? RG(slot, t) : make_float2(0.0f, 0.0f);
          slot = wrap1(slot + 1, C);
        }
        float xl[G], vl[G], ll[G], bden[G];
        xl[0] = xprev;
        vl[0] = vprev;
        ll[0] = llv;
#pragma unroll
        for (int u = 1; u < G; ++u) {
          xl[u] = cur[u - 1].x;
          vl[u] = cur[u - 1].y;
          ll[u] = d.car_l;
        }
        bool ok = t2_v_ok(vprev);
#pragma unroll
        for (int u = 0; u < G; ++u) {
          bden[u] = ((xl[u] - cur[u].x) - ll[u]) + d.eps;
          ok = ok && t2_v_ok(cur[u].y) && (__builtin_fabsf(bden[u]) >= TFX_T2_B_LO);
        }
        float xn[G], vn[G];
        if (fast_ok && __builtin_amdgcn_ballot_w64(!ok) == 0ull) {
#pragma unroll
          for (int u = 0; u < G; u += 2) {
            v2f x2, v2, vl2, b2, xo, vo;
            x2.x = cur[u].x; x2.y = cur[u + 1].x;
            v2.x = cur[u].y; v2.y = cur[u + 1].y;
            vl2.x = vl[u]; vl2.y = vl[u + 1];
            b2.x = bden[u]; b2.y = bden[u + 1];
            idm_pair(d, x2, v2, vl2, b2, xo, vo);
            xn[u] = xo.x; xn[u + 1] = xo.y;
            vn[u] = vo.x; vn[u + 1] = vo.y;
          }
        } else {
#pragma unroll
          for (int u = 0; u < G; ++u) idm_step(d, cur[u].x, cur[u].y, xl[u], vl[u], ll[u], xn[u], vn[u]);
        }
        xprev = cur[G - 1].x;  // OLD state leads the next car (Jacobi)
        vprev = cur[G - 1].y;
        llv = d.car_l;
#pragma unroll
        for (int u = 0; u < G; ++u) {
          const int k = my_k0 + k0 + u;  // the car's index on its road
          const bool act = k0 + u < my_n;
          if (act) RG(csl[u], t) = make_float2(xn[u], vn[u]);
          const bool pop = open && act && (xn[u] > d.length);  // the while loop of :123
          open = act ? pop : open;  // (rows past the lane's share leave it as it is: the halves combine on it)
          if (pop) {  // (rare: kept out of the common path - cfg1 x 1024: 8.67 -> 8.45 us per tick)
            ++kpop;
            far = far || ((xn[u] - d.length) > d.length);
          }
          const float wq = (k >= kq) ? xn[u] : vn[u];
          n_wait += (act && wq < d.thresh) ? 1 : 0;
          n_det += (act && xn[u] > d.near_end) ? 1 : 0;
          tail_x = act ? xn[u] : tail_x;
        }
      }
      if (LMAX > 1) {  // the shares meet in the road's first lane
        const int base = (int)(threadIdx.x & 63u & ~(unsigned)(lpr - 1));
        bool chain = open;  // every car so far left: the pop prefix runs on into the next share
#pragma unroll
        for (int j = 1; j < LMAX; ++j) {
          const int sj = base + (j < lpr ? j : 0);  // (every lane takes part in the exchange; a road with fewer lanes ignores it)
          const int kpop_b = __shfl(kpop, sj, 64), wait_b = __shfl(n_wait, sj, 64), det_b = __shfl(n_det, sj, 64);
          const int far_b = __shfl((int)far, sj, 64), n_b = __shfl(my_n, sj, 64), open_b = __shfl((int)open, sj, 64);
          const float tail_b = __shfl(tail_x, sj, 64);
          if (hA && j < lpr) {
            if (chain) {
              kpop += kpop_b;
              far = far || (far_b != 0);
            }
            chain = chain && (open_b != 0);
            n_wait += wait_b;
            n_det += det_b;
            if (n_b > 0) tail_x = tail_b;
          }
        }
      }
      if (run && hA) {
        if (train) {
          if (n_tot > 0) {
            wait_acc += n_wait;
            det = n_det;
            det_set = true;
          }
          s_passed[t] = (d.agent_mode && tt > 0) ? s_passed[t] + kpop : kpop;
          if (kpop > 0) s_pdst[li] = 1;
        }
        s_kpop[t] = kpop;
        s_lc[t] = lc;
        s_ovfsp[t] = ovf_sp;
        if (far) s_fartick[env_l] = tick + 1;
        if (kpop > 0) atomicMax(&s_maxpop[par], kpop);
        my_updates += (unsigned long long)n_tot;
      }
    }
    __syncthreads();  // B1: every ring holds its post-move cars, pops are published

    // ---- advance_finished_cars (:117-135) -------------------------------------------------------
    if (threadIdx.x == 0) s_maxpop[par ^ 1] = 0;
    const bool serial_env = valid && s_fartick[env_l] == tick + 1;
    const int rounds = (s_maxpop[par] + RES_KH - 1) / RES_KH;
    const bool pull = run && !serial_env && hA;
    const int ld_post = ring_adv(ld, kpop, C);
    const int k_p = (pull && pe >= 0) ? s_kpop[tp] : 0;
    const int ld_seen = (pe < e) ? ld : ld_post;  // p's pushes see leading[e] before e's own pops iff p < e
    int ovf = 0;
    if (serial_env && e == 0 && run && hA) {
      // literal single-lane loop for this env (a handed-off car is itself beyond the next road's end)
      // (q runs over ROAD IDS in the reference's order; COL(q) is where road q lives)
      float *rew = s_rew + (size_t)env_l * I;
      int overflowed = 0;
      if (!(d.accum_rewards && tt > 0))
        for (int i = 0; i < I; ++i) rew[i] = 0.0f;
      for (int q = 0; q < R; ++q) {  // spawn overflows happened before move_cars
        const int sp = s_ovfsp[COL(q)];
        if (sp > 0) {
          overflowed = 1;
          if (q < d.r)
            for (int j = 0; j < sp; ++j) rew[q % I] -= d.ovf_pen;
        }
      }
      for (int q = 0; q < d.r; ++q) s_passed[COL(q)] -= s_kpop[COL(q)];  // the parallel form's counts: recounted below
      for (int q = 0; q < R; ++q) {
        const int tq = COL(q);
        int l = s_ld[tq];
        while (l != s_lc[tq] && RG(wrap1(l + 1, C), tq).x > d.length) {
          const int newlead = wrap1(l + 1, C);
          const int nr = d.nexts[q];
          if (nr >= 0) {
            s_passed[tq] += 1;
            s_pdst[env_l * I + q % I] = 1;
            const float xc = RG(newlead, tq).x - d.length;
            const int tr = COL(nr);
            const int lcn = s_lc[tr], ldn = s_ld[tr];
            const int pos = wrap1(lcn + 1, C);
            const float start = (lcn != ldn) ? (RG(lcn, tr).x - d.car_l) - d.car_s0 : INFINITY;
            if (pos != ldn) {
              RG(pos, tr) = make_float2((start < xc) ? start : xc, RG(newlead, tq).y);
              if (W) RW(pos, tr) = RW(newlead, tq);
              s_lc[tr] = pos;
            } else {
              if (nr < d.r) rew[nr % I] -= d.ovf_pen;
              overflowed = 1;
            }
          }
          l = newlead;
          s_ld[tq] = l;
        }
      }
      for (int q = 0; q < R; ++q) {
        const int tq = COL(q);
        const int n = ring_count(s_ld[tq], s_lc[tq], C);
        s_cnt[tq] = n;
        s_tail[tq] = (n > 0) ? RG(s_lc[tq], tq).x : 0.0f;
        s_ovf[tq] = 0;
      }
      if (overflowed) {
        s_ovftick[env_l] = tick + 1;
        d.done_tick[env] = tick + 1;
      }
    }
    for (int rd = 0; rd < rounds; ++rd) {
      float2 car[RES_KH];
      float carw[RES_KH];
      int cnt = k_p - rd * RES_KH;
      cnt = cnt < 0 ? 0 : (cnt > RES_KH ? RES_KH : cnt);
      if (cnt > 0) {
        const int head_p = wrap1(s_ld[tp] + 1, C);
#pragma unroll
        for (int j = 0; j < RES_KH; ++j)
          if (j < cnt) {
            const int sp = ring_adv(head_p, rd * RES_KH + j, C);
            car[j] = RG(sp, tp);
            if (W) carw[j] = RW(sp, tp);
          }
      }
      __syncthreads();  // B2: the popped cars of this round are in registers everywhere
#pragma unroll
      for (int j = 0; j < RES_KH; ++j)
        if (j < cnt) {
          const float xc = car[j].x - d.length;  // state[e,xi,newlead] -= length (:130)
          const int pos = wrap1(lc + 1, C);
          const float start = (lc != ld_seen) ? (tail_x - d.car_l) - d.car_s0 : INFINITY;
          if (pos != ld_seen) {
            const float xv = (start < xc) ? start : xc;
            RG(pos, t) = make_float2(xv, car[j].y);
            if (W) RW(pos, t) = carw[j];
            lc = pos;
            tail_x = xv;
          } else {
            ++ovf;
          }
        }
    }
    if (pull) {
      const int n = ring_count(ld_post, lc, C);
      s_cnt[t] = n;
      s_tail[t] = tail_x;
      s_ovf[t] = ovf + ovf_sp;
      if (ovf + ovf_sp > 0) s_ovftick[env_l] = tick + 1;
    }
    __syncthreads();  // B3: every road of the pull form has read its predecessor's leading
    if (pull) {
      s_ld[t] = ld_post;
      s_lc[t] = lc;
    }
    // rewards[:] = 0 (:233) then -= OVERFLOW_PENALTY per dropped car (:110): exact in fp32
    if (des && run && !serial_env) {
      const int sum = s_ovf[c4[0]] + s_ovf[c4[1]] + s_ovf[c4[2]] + s_ovf[c4[3]];
      float rw = (d.accum_rewards && tt > 0) ? s_rew[li] : 0.0f;
      for (int j = 0; j < sum; ++j) rw -= d.ovf_pen;
      s_rew[li] = rw;
    }
    if (valid && e == 0 && hA && run && !serial_env && s_ovftick[env_l] == tick + 1) d.done_tick[env] = tick + 1;
    // (no barrier here: what the next tick reads before its first barrier - s_cnt / s_tail of the next
    // road, the light words of the other parity, the overflow stamps - was written before B3)
    if (++cyc_rem == cyc_P) { cyc_rem = 0; ++cyc_q; }
    if (++sp_rem == sp_P) sp_rem = 0;
    if (++gr_rem == gr_P) gr_rem = 0;
  }
  __syncthreads();

  // ---- store ------------------------------------------------------------------------------------
  if (valid) {
    ld = s_ld[t];
    lc = s_lc[t];
    const int n = s_cnt[t];
    const int head = wrap1(ld + 1, C);
    if (d.layout == 1) {
      const size_t col = tcol(d, env, e);
      for (int k = h; k < n; k += lpr) {
        const int slot = ring_adv(head, k, C);
        d.xv[col + (size_t)k * 64] = RG(slot, t);
        if (W) d.w[col + (size_t)k * 64] = RW(slot, t);
      }
      if (xL_set && hA) d.leadx[id] = xL;
    } else {
      for (int k = h; k < n; k += lpr) {
        const int slot = ring_adv(head, k, C);
        d.xv[(size_t)id * C + slot] = RG(slot, t);
        if (W) d.w[(size_t)id * C + slot] = RW(slot, t);
      }
      if (xL_set && hA) d.xv[(size_t)id * C + ld].x = xL;  // the fake leader's x sits in its slot (:133)
    }
  }
  if (valid && hA) {
    d.leading[id] = ld;
    d.lastcar[id] = lc;
    d.tailx[id] = s_tail[t];
    int *ob = d.obs + (size_t)env * d.obs_len;
    if (train) {
      ob[e] = s_passed[t];
      if (det_set) ob[d.r + e] = det;
      if (!a.tail) {
        if (wait_acc) d.waiting[(size_t)env * d.r + e] += wait_acc;
      } else {
        int *wp = d.waiting + (size_t)env * d.r + e;
        const int wtot = *wp + wait_acc;
        s_ovf[t] = wtot;                    // (free after the last tick) for the intersection's lane
        if (a.remi) *wp = 0;                // remi clears the counters it has read (:77)
        else if (wait_acc) *wp = wtot;
        if (a.aobs) {
          float *ao = a.aobs + (size_t)env * (2 * d.r + I);
          ao[e] = (float)s_passed[t];
          ao[d.r + e] = (float)(det_set ? det : ob[d.r + e]);
        }
      }
    }
    if (des) {
      const int2 pl = s_light[(size_t)((a.n_ticks - 1) & 1) * epb * I + li];
      ob[2 * d.r + isec] = pl.x;
      ob[2 * d.r + I + isec] = pl.y;
      if (!(a.tail && a.remi)) {
        d.rewards[(size_t)env * I + isec] = s_rew[li];
        if (s_pdst[li]) d.passed_dst[(size_t)env * I + isec] = 1;
      }
      if (a.greedy_spacing > 0) a.greedy_act[(size_t)env * I + isec] = s_act[li];
    }
    if (a.tail && a.adone && e == 0) a.adone[env] = s_ovftick[env_l] > tick0 ? 1 : 0;
  }
  if (a.tail) {
    __syncthreads();  // the roads' waiting totals are in s_ovf
    if (valid && des) {
      const int2 pl = s_light[(size_t)((a.n_ticks - 1) & 1) * epb * I + li];
      const size_t gi = (size_t)env * I + isec;
      float rw = s_rew[li];
      if (a.remi) {  // remi (:64-78) on the decision's final state, as k_remi does
        const bool pd = d.passed_dst[gi] != 0 || s_pdst[li] != 0;
        rw = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool green = ((q < 2) ? 1 : 0) != pl.x;
          const bool waiting = s_ovf[c4[q]] > 0;
          if (waiting && !green && !pd) rw -= 0.5f;
          else if (pd && green && !waiting) rw += 0.5f;
        }
        d.rewards[gi] = rw;
        d.passed_dst[gi] = 0;
      }
      if (a.areward) a.areward[gi] = rw;
      if (a.aobs)  // elapsed / 100 * (2 * phase - 1), computed in binary64 like the reference's NumPy expression
        a.aobs[(size_t)env * (2 * d.r + I) + 2 * d.r + isec] = (float)((double)pl.y / 100.0 * (double)(2 * pl.x - 1));
    }
  }
  if (a.poisson && (int)threadIdx.x < epb && (int)(blockIdx.x * epb + threadIdx.x) < d.E) {
    a.ps.gap_left[blockIdx.x * epb + threadIdx.x] = s_gap[threadIdx.x];
    a.ps.draws[blockIdx.x * epb + threadIdx.x] = s_draws[threadIdx.x];
  }
  for (int off = 32; off > 0; off >>= 1) my_updates += __shfl_down(my_updates, off);
  if ((threadIdx.x & 63) == 0 && my_updates) veh_add(d.veh, my_updates);
  if (a.own_clock && threadIdx.x == 0) {  // (one workgroup: nobody else reads the clock during this launch)
    *d.tickA = tick0 + a.n_ticks;
    *d.tickB = tick0 + a.n_ticks - 1;
  }
}

// the device clock after a resident launch (a separate launch: every workgroup of k_res reads tickA)
__global__ void k_tick_add(const Dev d, int n_ticks) {
  const int t = *d.tickA + n_ticks;
  *d.tickA = t;
  *d.tickB = t - 1;
}

}  // namespace tfx
