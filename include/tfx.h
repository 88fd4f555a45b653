/*
 * tfx.h - C ABI of the MI355X-native IDM traffic-env step ("tfx" = traffic step).
 *
 * This is the drop-in boundary for the hot path of samanklesaria/traffic-env.  The reference has
 * no FFI of its own: the boundary it offers is the set of numba-typed kernels in
 * gym_traffic/envs/traffic_env.py (each @jit signature is a C-like contract: typed, in-place,
 * caller-owned arrays) plus the TrafficEnv methods that sequence them.  Every entry point below
 * cites the reference interface it replaces.  All arrays gain a leading E (batched env)
 * dimension; E = 1 reproduces the reference object.
 *
 * Conventions
 *   - plain C, no torch/HIP types in signatures: device pointers are `void*`-compatible raw
 *     pointers, the stream is a `void*` holding a hipStream_t (NULL = default stream);
 *   - every call returns 0 on success, a negative TFX_E* code on failure; the message for the
 *     calling thread is available from tfx_last_error(); nothing throws, nothing calls exit();
 *   - all state arrays are owned by the caller (the Python env keeps them as PyTorch-ROCm
 *     tensors) and are mutated in place, exactly like the reference's NumPy arrays
 *     (traffic_env.py:361-382); the handle owns only static road tables and per-road scratch;
 *   - no global mutable state: handles are independent and may be driven from different host
 *     threads (the reference's kernels are nogil and A3C steps envs from threads, a3c.py:69-72).
 *
 * Device data layout (all little-endian, C-contiguous)
 *   xv         float32 [E][R][C][2]   (x, v) of the car in ring slot s of road e = the reference's
 *                                     state[e, xi, s], state[e, vi, s] (traffic_env.py:34,364),
 *                                     interleaved so a road's live cars are ONE contiguous span
 *                                     and a car is one 8-byte access.  16-byte aligned.
 *   w          float32 [E][R][C]      state[e, wi, s], the spawn tick (validate-mode trip times);
 *                                     present when tfx_config.planes == 3, else NULL.
 *                                     The other 7 per-car parameters are per-archetype constants
 *                                     (traffic_env.py:35-43) and live in tfx_config.
 *   leading    int32   [E][R]         slot of the fake leader  (README.md:14-23 of the reference)
 *   lastcar    int32   [E][R]         slot of the last car; == leading when the road is empty
 *   obs        int32   [E][2r+2I]     passed | detected | current_phase | elapsed (traffic_env.py:370-376)
 *   rewards    float32 [E][I]
 *   waiting    int32   [E][r]
 *   passed_dst uint8   [E][I]
 *   done_tick  int32   [E]            tick index (+1) of the last tick in which the env overflowed
 */
#ifndef TFX_H
#define TFX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFX_ABI_VERSION 11
#define TFX_KP 2 /* popped cars carried per road per tick on the parallel path; more -> exact serial path */
#define TFX_MAX_ARCH 64 /* rows of the archetype table (traffic_env.py:35-43 ships one); a power of two */

enum {
  TFX_OK = 0,
  TFX_EINVAL = -1,   /* bad argument / unsupported configuration */
  TFX_ESTATE = -2,   /* call order (e.g. step before bind_buffers) */
  TFX_EDEVICE = -3,  /* HIP runtime error (message has the HIP error string) */
  TFX_ENOMEM = -4
};

/* how the per-tick light action is obtained (TrafficEnv._step(action), traffic_env.py:224-232) */
enum {
  TFX_ACTION_BUFFER = 0,    /* int32 [n_ticks or 1][E][I] device buffer */
  TFX_ACTION_BROADCAST = 1, /* int32 [n_ticks or 1][I]: same action for every env */
  TFX_ACTION_CYCLE = 2,     /* on-device fixed cycle: a = ((tick + env % period) / period) & 1
                               (algorithms/fixed.py:6-7 with spacing = period, env-staggered) */
  TFX_ACTION_GREEDY = 3     /* on-device greedy controller (algorithms/greedy.py:14-16): every `period`
                               ticks phase 1 iff N-S approaches hold more cars than E-W ones */
};

/* how cars enter (TrafficEnv.add_new_cars, traffic_env.py:274-283) */
enum {
  TFX_SPAWN_NONE = 0,
  TFX_SPAWN_COUNTS = 1,  /* int32 [n_ticks or 1][E][n_entry]: cars to add to entry road j this tick
                            (host replays the reference's RandomState schedule) */
  TFX_SPAWN_PERIODIC = 2 /* on-device fixed rate: entry road e gets one car when
                            tick % period == e % period */
};

typedef struct tfx_config {
  int32_t m, n;          /* GridRoad(m, n, l): I = m*n, r = 4I, R = r + 2m + 2n (roadgraph.py:26-33) */
  int32_t capacity;      /* CAPACITY, slots per road incl. slot 0 and the fake leader (traffic_env.py:24) */
  int32_t n_envs;        /* E */
  int32_t planes;        /* 2: (x, v) only; 3: the w array is carried too */
  float length;          /* graph.len */
  float rate;            /* FLAGS.rate, seconds per tick (traffic_env.py:12) */
  /* the archetype, traffic_env.py:35-43 */
  float car_v, car_l, car_a, car_delta, car_v0, car_b, car_T, car_s0;
  int32_t yellow_ticks;  /* YELLOW_TICKS (traffic_env.py:21) */
  float thresh;          /* THRESH (traffic_env.py:17) */
  float detect_dist;     /* the 10 of `length - 10` (traffic_env.py:201) */
  float overflow_penalty;/* OVERFLOW_PENALTY (traffic_env.py:23) */
  float eps;             /* EPS (traffic_env.py:25) */
  int32_t learn_switch;  /* FLAGS.learn_switch (traffic_env.py:15,225-230) */
  int32_t validate;      /* FLAGS.mode == 'validate': advance_hack records trip times (traffic_env.py:240-242) */
  uint32_t entry_spec;   /* generate_entrypoints(spec) bit mask (roadgraph.py:42-51) */
  int32_t env_id_offset; /* global id of env 0 of this handle (env-sharded multi-GPU runs): the
                            on-device controllers use env + offset, so results do not depend on the
                            sharding */
  int32_t layout;        /* 0: xv is the ring layout above; 1: transposed layout - xv is T[tile][k][64][2]:
                            the k-th car behind the fake leader of the road in storage slot (64*tile + j)
                            at T[tile][k][j] (with planes = 3, w is T[tile][k][64] likewise);
                            tfx_xv_pairs gives the size, tfx_export_ring / tfx_import_ring convert to
                            and from the ring layout.  Between calls a column may start one or two rows
                            down (the handle remembers where: two-tick passes, tfx_pair_ticks), so T is
                            meaningful only together with its handle - snapshot, restore or edit the cars
                            through tfx_export_ring / tfx_import_ring, never through T itself */
  /* The reference's `archetypes` TABLE (traffic_env.py:35-43: float32 [n][10]; add_new_cars draws a row per car,
   * :164).  n_archetypes <= 1: the single archetype of the car_* fields above (the reference's default), and the
   * table is ignored.  n_archetypes in 2..TFX_MAX_ARCH, or one row whose delta is not 4: "heterogeneous cars" -
   * every car carries the row it was spawned from through handoffs; needs layout = 1 and planes = 3 (the per-car
   * side word then holds (spawn tick mod 2^24) << 6 | row as integer bits; the envs never run LDS-resident), and takes the rows of spawned
   * cars from tfx_set_spawn_archetypes.  (v/v0)**delta: for an integer delta in 1..8 the binary64 product chain of
   * oracle/idm_oracle.c powi_cr rounded once (for 4: the same value as the single-archetype path); for any other delta
   * in (0, 64] the binary64 log2 / exp2 sequence of include/tfx_pow.h rounded once - both shared bit for bit with the
   * oracle, both within 1 ulp of NumPy's float32 power, which is what the reference runs (traffic_env.py:56).
   * Row layout: v (spawn speed), l, a, delta, v0, b, T, s0. */
  int32_t n_archetypes;
  float arch[TFX_MAX_ARCH][8];
} tfx_config;

typedef struct tfx_buffers {
  float *xv;
  float *w;
  int32_t *leading;
  int32_t *lastcar;
  int32_t *obs;
  float *rewards;
  int32_t *waiting;
  uint8_t *passed_dst;
  int32_t *done_tick;
  float *trip_times;   /* [E][trip_cap] or NULL; (tick - w)/2 of cars leaving the map (traffic_env.py:154) */
  int32_t *n_trips;    /* [E] or NULL */
  int32_t trip_cap;
} tfx_buffers;

typedef struct tfx_handle_s *tfx_handle;

int tfx_abi_version(void);
const char *tfx_last_error(void);

/* TrafficEnv.set_graph (traffic_env.py:361-382) + GridRoad tables (roadgraph.py:26-64): builds
 * dest/phases/nexts/entrypoints for the grid on the device.  */
int tfx_create(const tfx_config *cfg, tfx_handle *out);
int tfx_destroy(tfx_handle h);
/* sizes for the caller's allocations */
int tfx_dims(tfx_handle h, int32_t *I, int32_t *r, int32_t *R, int32_t *n_entry);
/* copies the int32[R] tables (host pointers, any may be NULL); entrypoints is int32[n_entry] */
int tfx_tables(tfx_handle h, int32_t *dest, int32_t *phases, int32_t *nexts, int32_t *entrypoints);
int tfx_bind_buffers(tfx_handle h, const tfx_buffers *b);

/* TrafficEnv._reset (traffic_env.py:259-272).  phase_init: device int32 [E][I] (replaces
 * action_space.sample()).  detected / rewards are left stale, as in the reference. */
int tfx_reset(tfx_handle h, const int32_t *phase_init, void *stream);
/* The batched form's episode boundary: _reset for the envs whose byte in `mask` (device uint8 [E]) is
 * non-zero, the others untouched; phase_init (device int32 [E][I]) is read for those envs only.  The
 * device clock is shared by the batch and keeps running (the reference's `steps` only feeds the
 * spawn-tick stamps, whose differences are what trip times use). */
int tfx_reset_envs(tfx_handle h, const int32_t *phase_init, const uint8_t *mask, void *stream);
/* Call after writing state/leading/lastcar from outside (tests, checkpoint restore): rebuilds the
 * per-road tail cache the light kernel reads. */
int tfx_refresh(tfx_handle h, void *stream);

int tfx_set_actions(tfx_handle h, int32_t mode, const int32_t *dev, int32_t period, int32_t per_tick);
int tfx_set_spawns(tfx_handle h, int32_t mode, const int32_t *dev, int32_t period, int32_t per_tick);
/* On-device form of the reference's Poisson generator (traffic_env.py:160-164): per env, gaps of
 * round(Exp(1/cars_per_tick)) ticks between cars, each car on a uniformly drawn entry road; Philox
 * streams keyed by (seed, global env id).  `cdf` (host pointer, n_cdf entries) holds
 * P(gap <= k) * 2^32 for k = 0.. (the last entry must be 0xFFFFFFFF); gym_traffic/devrng.py builds it
 * and mirrors the stream on the host. */
int tfx_set_poisson(tfx_handle h, double cars_per_tick, uint64_t seed, const uint32_t *cdf, int32_t n_cdf);
/* On-device form of the reference's `regular` generator (traffic_env.py:167-176): with cars_per_tick =
 * cars_per_sec * rate, `burst` = ceil(cars_per_tick) cars in every tick i of the env's generator with
 * i % every == 0, `every` = round(1 / cars_per_tick) (Python's round: half to even; every == 0 means every tick) -
 * the caller passes the two integers; each car on a uniformly drawn entry road (rand.choice(entrypoints), :280) from the
 * Philox stream keyed by (seed, global env id), car c using the same draw index as car c of tfx_set_poisson's stream.
 * The per-tick car COUNTS are exactly the reference's; gym_traffic/devrng.py mirrors the road draws on the host. */
int tfx_set_regular(tfx_handle h, int32_t every, int32_t burst, uint64_t seed);
/* Heterogeneous cars only: the archetype row of every car the count buffer of tfx_set_spawns adds
 * (`archetypes[random.randint(archetypes.shape[0])]`, traffic_env.py:164): device uint8
 * [n_ticks or 1][E][n_entry][per_road], entry j of a road = its j-th car of the tick in creation order (cars
 * beyond per_road, and every car while no buffer is bound or under TFX_SPAWN_PERIODIC - the reference's `regular`
 * generator yields archetypes[0], :174 - get row 0). */
int tfx_set_spawn_archetypes(tfx_handle h, const uint8_t *dev, int32_t per_road, int32_t per_tick);

/* TrafficEnv._step (traffic_env.py:224-248), n_ticks times: phase/elapsed update, spawns,
 * move_cars, advance_finished_cars | advance_hack, steps += 1. */
int tfx_step(tfx_handle h, int32_t n_ticks, void *stream);
/* The two halves on their own, for kernel-level parity tests:
 * move_cars (traffic_env.py:187-212, incl. update_lights :81-94; also applies the phase update and
 * the spawns of the current tick) and advance_finished_cars / advance_hack (:117-157). */
int tfx_move_cars(tfx_handle h, void *stream);
int tfx_advance_finished_cars(tfx_handle h, void *stream);

/* One agent decision = the Repeater (+ Remi) wrappers of traffic_test.py:27-64 fused on the device:
 * n_ticks x _step with the held action; `passed` accumulates, `detected` keeps the last tick, an
 * env that overflows stops for the rest of the step (`if done: break`); then, with remi != 0,
 * remi_reward() (else the rewards are the sum over the ticks).  Outputs (device pointers, any may
 * be NULL): aobs float32 [E][2r+I] = [sum passed | last detected | elapsed/100*(2*phase-1)],
 * areward float32 [E][I], adone uint8 [E].  The launch sequence is captured into a HIP graph on
 * first use and replayed afterwards.  Needs ONE action for the whole step (a held buffer, the cycle
 * rule or the greedy controller); any spawn rule works, a per-tick count buffer must hold at least
 * n_ticks rows (row t feeds tick t of the step; rows after an env's overflow are not consumed). */
int tfx_agent_step(tfx_handle h, int32_t n_ticks, int32_t remi, float *aobs, float *areward,
                   uint8_t *adone, void *stream);

/* remi (traffic_env.py:64-78) via TrafficEnv.remi_reward (:384-387) */
int tfx_remi(tfx_handle h, void *stream);
/* cars_on_roads (traffic_env.py:214-218): out device int32 [E][R] */
int tfx_cars_on_roads(tfx_handle h, int32_t *out, void *stream);
/* done flag of _step (traffic_env.py:246-248): out[k] = env k overflowed in a tick >= since_tick */
int tfx_done(tfx_handle h, uint8_t *out, int32_t since_tick, void *stream);

int tfx_get_tick(tfx_handle h, int32_t *tick);              /* TrafficEnv.steps */
int tfx_set_tick(tfx_handle h, int32_t tick);               /* also clears the per-env overflow stamps */
/* live cars advanced by move_cars since the last tfx_reset_counters (synchronises the stream) */
int tfx_vehicle_updates(tfx_handle h, uint64_t *out, void *stream);
int tfx_reset_counters(tfx_handle h, void *stream);
/* Per-kernel timing for the roofline report: with max_ticks > 0 the next tfx_step calls record HIP
 * events on the launch stream around the move and advance kernels of up to max_ticks ticks;
 * tfx_profile_read waits for them and returns (and clears) the summed durations. 0 disables. */
int tfx_profile(tfx_handle h, int32_t max_ticks);
int tfx_profile_read(tfx_handle h, double *move_ms, double *advance_ms, int32_t *n_ticks);
/* (x, v) pairs the caller's xv buffer must hold for this handle's layout (and, with planes = 3 on the
 * transposed layout, floats its w buffer must hold) */
int tfx_xv_pairs(tfx_handle h, int64_t *pairs);
/* Transposed-layout handles: copy the cars to / from ring-layout arrays (device pointers): ring_xv
 * float32 [E][R][C][2] with the fake leader's x in slot `leading` as the reference keeps it, ring_w
 * float32 [E][R][C] (spawn ticks; may be NULL, ignored unless planes = 3), ring_a uint8 [E][R][C] (archetype
 * row per car; may be NULL, ignored unless the handle has heterogeneous cars).  After an import call
 * tfx_refresh. */
int tfx_export_ring(tfx_handle h, float *ring_xv, float *ring_w, uint8_t *ring_a, void *stream);
int tfx_import_ring(tfx_handle h, const float *ring_xv, const float *ring_w, const uint8_t *ring_a, void *stream);

/* Two of the IDM's three divisions have a constant divisor (2*sqrt(a*b) and v0).  At tfx_create the
 * library checks on the device, exhaustively over the admitted numerator range, that the
 * reciprocal form it would like to use is bit-identical to IEEE division for these constants;
 * `enabled` reports whether it is in use, `mismatches` the count found (0 when enabled). */
int tfx_fastdiv_status(tfx_handle h, int32_t *enabled, uint64_t *mismatches);
/* launch geometry of the move kernel, for the roofline report */
int tfx_launch_info(tfx_handle h, int32_t *grid, int32_t *block, int32_t *waves_per_road);
/* Ticks of this handle that ran in the LDS-resident multi-tick kernel (k_res: every tick of a tfx_step
 * or tfx_agent_step call in ONE launch, the envs' cars held in a compute unit's LDS) since tfx_create,
 * and whether the handle's envs fit it at all (`capable`: two lanes per road - or one - within 512
 * lanes and 160 KB of rings per workgroup: cfg0, cfg1, the reference's 3x3 default do; cfg2 and cfg4
 * do not).  A capable handle takes k_res on its own unless trip times are recorded (validate mode);
 * every input rule runs inside it (held / per-tick buffers, fixed cycle, periodic arrivals, the
 * on-device Poisson stream, the greedy controller) and tfx_agent_step's remi / observation / done tail
 * too; results are bit-identical either way.  Environment switches read by tfx_bind_buffers:
 * TFX_RESIDENT=0 turns it off, TFX_RES_LPR=1 forces one lane per road, TFX_RES_EPB=n packs n envs per
 * workgroup, TFX_RES_MIN_TICKS=n leaves calls shorter than n ticks to the per-tick kernels. */
int tfx_fused_ticks(tfx_handle h, int64_t *ticks, int32_t *capable);
/* Ticks of this handle that ran as two-tick passes since tfx_create (transposed layout: k_move_tt takes every
 * car but the head of each road through TWO ticks per trip through HBM, k_edge finishes the second tick for
 * the heads and the cars that joined a road in between - csrc/tfx_move_tt.hpp; launches that leave wave slots empty
 * split every tile's walk over 2, 4 or 8 wavefronts: k_move_tts, csrc/tfx_move_tts.hpp).  tfx_step and tfx_agent_step
 * use them on their own for calls of two ticks or more of every handle whose envs do not fit k_res - heterogeneous
 * cars from 4 tiles of 64 roads per compute unit on; results are
 * bit-identical to the tick-by-tick kernels.  TFX_PAIRS=0 turns them off, TFX_PAIRS=2 forces them at any size. */
int tfx_pair_ticks(tfx_handle h, int64_t *ticks);
/* ... of which the rest of the pair - advance_finished_cars of the first tick (traffic_env.py:117-135), the road
 * heads' second tick, advance_finished_cars of the second - ran as ONE launch with a workgroup per env (k_tail,
 * csrc/tfx_tail.hpp) instead of three: tfx_step does so on its own from one env per compute unit on, unless the
 * second tick's inputs are produced on the device in between (tfx_set_poisson, TFX_ACTION_GREEDY).  TFX_TAIL=0
 * turns it off, TFX_TAIL=2 forces it at any batch size; results are bit-identical. */
int tfx_tail_ticks(tfx_handle h, int64_t *ticks);
/* Inside tfx_agent_step a pair of ticks runs as ONE pass over the cars only for envs in which the pair's first tick
 * provably cannot overflow a ring (a bound on how far a car can move, csrc/tfx_move_tt.hpp risk_lane); the others take
 * the pair one tick at a time, which is what `if done: break` (traffic_test.py:55) needs.  pairs = env-pairs that took
 * that slower, equally exact path since tfx_create (synchronises the stream). */
int tfx_slow_pairs(tfx_handle h, uint64_t *pairs, void *stream);
/* Ticks of tfx_step calls that ran as two halves of the env range, the second half on a stream the handle owns
 * (forked from and joined to the caller's stream with events, so the call keeps its stream semantics): the
 * latency-bound per-road launch of one half then runs under the other half's pass over the cars.  Used for calls
 * of pairs whose halves still fill the chip, never while tfx_profile is timing kernels.  TFX_SPLIT=0 turns it off,
 * TFX_SPLIT=2 forces it at any batch size; results are bit-identical (envs share nothing, traffic_env.py:361-382). */
int tfx_split_ticks(tfx_handle h, int64_t *ticks);
/* name of the kernel that moved the cars in the handle's last tick ("k_move_tt", "k_move_tts", "k_move_t", "k_move_ts",
 * "k_res", "k_move_dma", ...), for the roofline report; "" before the first step */
const char *tfx_step_kernel(tfx_handle h);

/* Error-path testing: the n-th kernel launch a later tfx_step / tfx_agent_step / tfx_move_cars /
 * tfx_advance_finished_cars call would make (counted from this call, over calls) is not made and that call returns
 * TFX_EDEVICE instead.  The handle stays usable: its second stream is joined back, nothing a sequence changes in the
 * handle while it enqueues stays changed; the envs' state is then somewhere inside the failed call (reset or reload
 * it).  0 switches the injection off. */
int tfx_debug_fail_after(tfx_handle h, int32_t n_launches);

/* Host-side replay of the reference's seeded arrival generators for many envs (no GPU involved): one
 * stream per env holds a legacy numpy RandomState's MT19937 state (`RandomState.get_state()[1:3]`)
 * plus the generator's own state - `gap`: whole ticks left before the next Poisson car (-1 = draw a
 * new gap), `tick`: the regular generator's tick counter.  tfx_arrivals_replay advances every stream
 * by n_ticks exactly as traffic_env.py:160-176 + :274-283 would (poisson != 0: round(Exp(mean_gap))
 * empty ticks between cars; else `burst` cars every `every` ticks), drawing each car's entry road
 * with rand.choice over n_choices entry points.  counts int32 [n_ticks][n_streams][n_columns] receives
 * the cars per entry road (choice c -> column column_of_choice[c]), made int32 [n_ticks][n_streams]
 * (may be NULL) the cars created. */
typedef struct tfx_arrival_stream {
  uint32_t mt[624];
  int32_t pos;
  int32_t gap;
  int64_t tick;
} tfx_arrival_stream;
int tfx_arrivals_replay(tfx_arrival_stream *streams, int32_t n_streams, int32_t n_ticks, int32_t poisson,
                        double mean_gap, int32_t every, int32_t burst, int32_t n_choices,
                        const int32_t *column_of_choice, int32_t n_columns, int32_t *counts, int32_t *made);

#ifdef __cplusplus
}
#endif
#endif /* TFX_H */
