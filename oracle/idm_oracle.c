/*
 * idm_oracle.c - CPU restatement of the reference's IDM env tick.  TEST INFRASTRUCTURE.
 *
 * This file is the parity ORACLE for the HIP path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product package never does.
 *
 * It restates, function for function and in the reference's own sequential order, the
 * numba/NumPy kernels of /root/reference/gym_traffic/envs/traffic_env.py (cited per function
 * below).  It keeps the reference's full state layout float32[R][10][C] (param-major per
 * road, traffic_env.py:33-34,364) so that every read the reference makes (leader length,
 * leader speed, per-car a/b/T/s0/v0/delta) is the same read here.
 *
 * Floating-point contract, shared bit-for-bit with the HIP kernels:
 *   - every operation is IEEE-754 binary32, evaluated in the reference's expression order,
 *     one rounding per operation (build with -ffp-contract=off, no fast-math);
 *   - sqrt and divide are correctly rounded;
 *   - (v/v0)**delta for delta == 4 (the only archetype the reference defines,
 *     traffic_env.py:38) is q^4 rounded ONCE to binary32, computed as
 *     (float)(((double)q*q)*((double)q*q)).  NumPy's float32 array power is a platform SIMD
 *     routine within 1 ulp of that (equal for ~79 % of inputs on the capture host), so golden
 *     floats captured from the reference differ from this contract by <= 1 ulp of the power
 *     term per tick - the resulting tolerance is stated in the tests;
 *   - np.maximum(0, t) is restated as (0 >= t) ? 0 : t  (NaN propagates, as in NumPy);
 *   - (dx > 0) * dx is restated as dx > 0 ? dx : 0.0f * dx (keeps NumPy's NaN for dx = -inf).
 *
 * Pinning: tests/test_oracle_golden.py checks this file against the golden vectors captured
 * from the reference itself (oracle/gen_golden.py).
 */
#include "../include/tfx_pow.h"
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define NPARAMS 10
enum { XI = 0, VI, LI, AI, DELTAI, V0I, BI, TI, S0I, WI }; /* traffic_env.py:34 */

typedef struct orc_cfg {
  int32_t m, n;            /* grid */
  int32_t I, r, R;         /* intersections, train roads, roads (roadgraph.py:30-33) */
  int32_t C;               /* CAPACITY: slots per road (traffic_env.py:24) */
  int32_t E;               /* batched env count (the reference has E = 1) */
  float length;            /* graph.len */
  float rate;              /* FLAGS.rate (traffic_env.py:12) */
  float archetype[NPARAMS]; /* traffic_env.py:35-43 */
  int32_t yellow_ticks;    /* YELLOW_TICKS = 6   (traffic_env.py:21) */
  float thresh;            /* THRESH = 0.2       (traffic_env.py:17) */
  float detect_dist;       /* 10 in `length - 10` (traffic_env.py:201) */
  float overflow_penalty;  /* OVERFLOW_PENALTY = 10 (traffic_env.py:23) */
  float eps;               /* EPS = 1e-8         (traffic_env.py:25) */
  int32_t learn_switch;    /* FLAGS.learn_switch (traffic_env.py:15) */
  int32_t validate;        /* FLAGS.mode == 'validate' (traffic_env.py:240) */
} orc_cfg;

/* Per-env views into the batched arrays. */
typedef struct orc_env {
  float *state;        /* [R][10][C] */
  int32_t *leading;    /* [R] */
  int32_t *lastcar;    /* [R] */
  int32_t *obs;        /* [2r+2I] = passed | detected | current_phase | elapsed (traffic_env.py:370-376) */
  float *rewards;      /* [I] */
  int32_t *waiting;    /* [r] */
  uint8_t *passed_dst; /* [I] */
} orc_env;

#define ST(env, cfg, e, p, s) ((env)->state[((size_t)(e) * NPARAMS + (p)) * (cfg)->C + (s)])

/* traffic_env.py:46-47  wrap */
static inline int32_t wrap(const orc_cfg *c, int32_t a) { return a >= c->C ? 1 : a; }

static inline float np_max0(float t) { return (0.0f >= t) ? 0.0f : t; }

/* q**4 rounded once from the binary64 product: q*q is exact in binary64, the square of that
 * carries one binary64 rounding, so the binary32 result is the correctly rounded q^4 except
 * for double-rounding ties (~2^-29 of inputs).  Same two binary64 multiplies on the GPU. */
static inline float pow4_cr(float q) {
  const double q2 = (double)q * (double)q;
  return (float)(q2 * q2);
}

/* (v/v0)**delta for the other integer exponents an archetype may carry (traffic_env.py:38 fixes 4; a user row may
 * not): binary exponentiation in binary64 - result = 1, base = q; while n: if n odd result *= base; base *= base;
 * n >>= 1 - rounded once to binary32.  For n = 4 that is (q*q)*(q*q), i.e. pow4_cr.  The HIP side runs the same
 * multiplies; any other exponent goes through tfx_pow_det (include/tfx_pow.h): one binary64 operation sequence that
 * both sides compile, rounded once to binary32. */
static inline float powi_cr(float q, int n) {
  double result = 1.0, base = (double)q;
  while (n) {
    if (n & 1) result *= base;
    base *= base;
    n >>= 1;
  }
  return (float)result;
}

/*
 * traffic_env.py:50-62  sim(r, ld, me): IDM over the slot range me = [lo+1 .. hi], ld = [lo .. hi-1]
 * of road e.  All temporaries are formed from the OLD values (NumPy evaluates whole arrays
 * before the two in-place writes), hence the two passes.
 */
static void sim(const orc_cfg *c, orc_env *v, int e, int lo, int hi, float *nx, float *nv) {
  const float r = c->rate;
  for (int j = lo + 1; j <= hi; ++j) {
    const float vel = ST(v, c, e, VI, j);
    const float ldv = ST(v, c, e, VI, j - 1);
    const float t_gap = vel * ST(v, c, e, TI, j);
    const float appr = vel * (vel - ldv);
    const float two_sab = 2.0f * sqrtf(ST(v, c, e, AI, j) * ST(v, c, e, BI, j));
    const float s_star = ST(v, c, e, S0I, j) + np_max0(t_gap + appr / two_sab);
    const float s = ST(v, c, e, XI, j - 1) - ST(v, c, e, XI, j) - ST(v, c, e, LI, j - 1);
    const float q = vel / ST(v, c, e, V0I, j);
    const float delta = ST(v, c, e, DELTAI, j);
    const float qd = (delta == 4.0f) ? pow4_cr(q)
                     : (delta >= 1.0f && delta <= 8.0f && delta == (float)(int)delta) ? powi_cr(q, (int)delta)
                                                                                     : tfx_pow_det(q, delta);
    const float u = s_star / (s + c->eps);
    const float dv = ST(v, c, e, AI, j) * ((1.0f - qd) - u * u);
    const float dvr = dv * r;
    const float dx = r * vel + (0.5f * dvr) * r;
    nx[j] = ST(v, c, e, XI, j) + (dx > 0.0f ? dx : 0.0f * dx);
    nv[j] = np_max0(vel + dvr);
  }
  for (int j = lo + 1; j <= hi; ++j) {
    ST(v, c, e, XI, j) = nx[j];
    ST(v, c, e, VI, j) = nv[j];
  }
}

/* traffic_env.py:81-94  update_lights */
static void update_lights(const orc_cfg *c, const int32_t *dests, const int32_t *phases,
                          const int32_t *nexts, orc_env *v) {
  const int32_t *cur = v->obs + 2 * c->r;
  const int32_t *elapsed = v->obs + 2 * c->r + c->I;
  for (int e = 0; e < c->R; ++e) {
    const int dst = dests[e];
    if (dst == -1) return;
    if (phases[e] == cur[dst] || elapsed[dst] < c->yellow_ticks) {
      ST(v, c, e, XI, v->leading[e]) = c->length;
    } else {
      const int nr = nexts[e];
      if (nr >= 0 && v->lastcar[nr] != v->leading[nr]) {
        float t = ST(v, c, nr, XI, v->lastcar[nr]);
        t += c->length;
        ST(v, c, e, XI, v->leading[e]) = t;
      } else {
        ST(v, c, e, XI, v->leading[e]) = INFINITY;
      }
    }
  }
}

/* traffic_env.py:187-212  move_cars.  Returns the number of live cars advanced (the
 * "vehicle-updates" of this tick; bookkeeping for the benchmark, not part of the reference). */
int64_t orc_move_cars(const orc_cfg *c, const int32_t *dests, const int32_t *phases,
                      const int32_t *nexts, orc_env *v) {
  const int C = c->C;
  float nx[C + 1], nv[C + 1];
  int32_t *detected = v->obs + c->r;
  const float near_end = c->length - c->detect_dist;
  int64_t updates = 0;
  update_lights(c, dests, phases, nexts, v);
  for (int e = 0; e < c->R; ++e) {
    const int ld = v->leading[e], lc = v->lastcar[e];
    if (ld == lc) continue;
    updates += lc - ld + (ld > lc ? C - 1 : 0);
    if (ld < lc) {
      sim(c, v, e, ld, lc, nx, nv);
      if (dests[e] >= 0) {
        int w = 0, d = 0;
        for (int j = ld + 1; j <= lc; ++j) {
          w += ST(v, c, e, VI, j) < c->thresh;
          d += ST(v, c, e, XI, j) > near_end;
        }
        v->waiting[e] += w;
        detected[e] = d;
      }
    } else {
      for (int p = 0; p < NPARAMS; ++p) ST(v, c, e, p, 0) = ST(v, c, e, p, C - 1);
      sim(c, v, e, ld, C - 1, nx, nv);
      sim(c, v, e, 0, lc, nx, nv);
      if (dests[e] >= 0) {
        int w = 0, d = 0;
        for (int j = ld + 1; j <= C - 1; ++j) {
          w += ST(v, c, e, VI, j) < c->thresh;
          d += ST(v, c, e, XI, j) > near_end;
        }
        /* the reference counts x (not v) below THRESH on the second segment (:210) - kept */
        for (int j = 1; j <= lc; ++j) {
          w += ST(v, c, e, XI, j) < c->thresh;
          d += ST(v, c, e, XI, j) > near_end;
        }
        v->waiting[e] += w;
        detected[e] = d;
      }
    }
  }
  return updates;
}

/* traffic_env.py:97-114  add_car; `car` is a 10-vector */
static int add_car(const orc_cfg *c, int road, const float *car, orc_env *v, const int32_t *dests) {
  const int pos = wrap(c, v->lastcar[road] + 1);
  float start = INFINITY;
  if (v->lastcar[road] != v->leading[road]) {
    const int t = v->lastcar[road];
    start = ST(v, c, road, XI, t) - ST(v, c, road, LI, t) - ST(v, c, road, S0I, t);
  }
  if (pos != v->leading[road]) {
    for (int p = 0; p < NPARAMS; ++p) ST(v, c, road, p, pos) = car[p];
    const float x = ST(v, c, road, XI, pos);
    ST(v, c, road, XI, pos) = (start < x) ? start : x; /* python min(x, start) */
    v->lastcar[road] = pos;
    return 0;
  }
  if (dests[road] >= 0) v->rewards[dests[road]] -= c->overflow_penalty;
  return 1;
}

/*
 * traffic_env.py:117-135 advance_finished_cars and :139-157 advance_hack (validate != 0).
 * trip_times (may be NULL) receives (tick - w)/2 for cars leaving the map; *n_trips is advanced,
 * at most trip_cap entries are stored.
 */
int orc_advance(const orc_cfg *c, const int32_t *dests, const int32_t *nexts, orc_env *v,
                float tick, float *trip_times, int64_t *n_trips, int64_t trip_cap) {
  int overflowed = 0;
  int32_t *passed = v->obs;
  float car[NPARAMS];
  for (int e = 0; e < c->R; ++e) {
    while (v->leading[e] != v->lastcar[e] &&
           ST(v, c, e, XI, wrap(c, v->leading[e] + 1)) > c->length) {
      const int newlead = wrap(c, v->leading[e] + 1);
      const int nr = nexts[e];
      if (nr >= 0) {
        passed[e] += 1;
        v->passed_dst[dests[e]] = 1;
        ST(v, c, e, XI, newlead) -= c->length;
        for (int p = 0; p < NPARAMS; ++p) car[p] = ST(v, c, e, p, newlead);
        overflowed = add_car(c, nr, car, v, dests) || overflowed;
      } else if (c->validate) {
        if (n_trips) {
          if (trip_times && *n_trips < trip_cap) trip_times[*n_trips] = (tick - ST(v, c, e, WI, newlead)) / 2.0f;
          *n_trips += 1;
        }
      }
      for (int p = 0; p < NPARAMS; ++p) ST(v, c, e, p, newlead) = ST(v, c, e, p, v->leading[e]);
      v->leading[e] = newlead;
    }
  }
  return overflowed;
}

/* traffic_env.py:64-78  remi */
void orc_remi(const orc_cfg *c, const int32_t *dests, const int32_t *phases, orc_env *v) {
  const int32_t *cur = v->obs + 2 * c->r;
  for (int i = 0; i < c->I; ++i) v->rewards[i] = 0.0f;
  for (int e = 0; e < c->R; ++e) {
    const int dst = dests[e];
    if (dst == -1) break;
    const int green = phases[e] != cur[dst];
    if (v->waiting[e] > 0 && !green && !v->passed_dst[dst]) v->rewards[dst] -= 0.5f;
    else if (v->passed_dst[dst] && green && !(v->waiting[e] > 0)) v->rewards[dst] += 0.5f;
  }
  memset(v->passed_dst, 0, (size_t)c->I);
  for (int e = 0; e < c->r; ++e) v->waiting[e] = 0;
}

/* traffic_env.py:214-218  cars_on_roads */
void orc_cars_on_roads(const orc_cfg *c, const orc_env *v, int32_t *out) {
  for (int e = 0; e < c->R; ++e) {
    const int inverted = v->leading[e] > v->lastcar[e];
    out[e] = inverted * (c->C - 1) + v->lastcar[e] - v->leading[e];
  }
}

/* traffic_env.py:259-272  _reset (phase_init replaces action_space.sample()) */
void orc_reset(const orc_cfg *c, orc_env *v, const int32_t *phase_init) {
  for (int e = 0; e < c->R; ++e) {
    for (int p = 0; p < NPARAMS; ++p) ST(v, c, e, p, 1) = 0.0f;
    ST(v, c, e, XI, 1) = INFINITY;
    v->leading[e] = 1;
    v->lastcar[e] = 1;
  }
  int32_t *passed = v->obs, *cur = v->obs + 2 * c->r, *elapsed = cur + c->I;
  for (int i = 0; i < c->I; ++i) { elapsed[i] = 0; v->passed_dst[i] = 0; cur[i] = phase_init[i]; }
  for (int e = 0; e < c->r; ++e) { passed[e] = 0; v->waiting[e] = 0; }
}

/*
 * traffic_env.py:224-248  _step for ONE env.  `action` int32[I]; `spawn_roads` lists the entry
 * road of every car add_new_cars (:274-283) would create this tick, in order.  Returns overflowed.
 */
int orc_step_arch(const orc_cfg *c, const int32_t *dests, const int32_t *phases, const int32_t *nexts,
                  orc_env *v, const int32_t *action, const int32_t *spawn_roads, int n_spawn,
                  float tick, float *trip_times, int64_t *n_trips, int64_t trip_cap, int64_t *updates,
                  const float *archetypes, const int32_t *spawn_arch);

int orc_step(const orc_cfg *c, const int32_t *dests, const int32_t *phases, const int32_t *nexts,
             orc_env *v, const int32_t *action, const int32_t *spawn_roads, int n_spawn,
             float tick, float *trip_times, int64_t *n_trips, int64_t trip_cap, int64_t *updates) {
  return orc_step_arch(c, dests, phases, nexts, v, action, spawn_roads, n_spawn, tick, trip_times, n_trips, trip_cap,
                       updates, NULL, NULL);
}

/* ... with the reference's `archetypes` table (traffic_env.py:35-43: float32[n][10], any number of rows) and, per
 * spawned car, the row add_new_cars drew for it (`archetypes[random.randint(archetypes.shape[0])]`, :164); NULL
 * table = the single archetype of the config. */
int orc_step_arch(const orc_cfg *c, const int32_t *dests, const int32_t *phases, const int32_t *nexts,
                  orc_env *v, const int32_t *action, const int32_t *spawn_roads, int n_spawn,
                  float tick, float *trip_times, int64_t *n_trips, int64_t trip_cap, int64_t *updates,
                  const float *archetypes, const int32_t *spawn_arch) {
  int32_t *passed = v->obs, *cur = v->obs + 2 * c->r, *elapsed = cur + c->I;
  for (int i = 0; i < c->I; ++i) {
    int change;
    if (c->learn_switch) {
      change = action[i] != 0;
      cur[i] = (cur[i] != 0) != (action[i] != 0);
    } else {
      change = (cur[i] != 0) != (action[i] != 0);
      cur[i] = action[i];
    }
    elapsed[i] += 1;
    elapsed[i] *= !change;
  }
  for (int i = 0; i < c->I; ++i) v->rewards[i] = 0.0f;
  for (int e = 0; e < c->r; ++e) passed[e] = 0;
  int overflowed = 0;
  float car[NPARAMS];
  for (int k = 0; k < n_spawn; ++k) {
    memcpy(car, archetypes ? archetypes + (size_t)(spawn_arch ? spawn_arch[k] : 0) * NPARAMS : c->archetype, sizeof car);
    car[WI] = tick;
    overflowed = add_car(c, spawn_roads[k], car, v, dests) || overflowed;
  }
  const int64_t u = orc_move_cars(c, dests, phases, nexts, v);
  if (updates) *updates += u;
  overflowed = orc_advance(c, dests, nexts, v, tick, trip_times, n_trips, trip_cap) || overflowed;
  return overflowed;
}

/* ---- batched entry points (ctypes) ------------------------------------------------------- */

typedef struct orc_bufs {
  float *state; int32_t *leading; int32_t *lastcar; int32_t *obs; float *rewards;
  int32_t *waiting; uint8_t *passed_dst; uint8_t *done;
} orc_bufs;

static orc_env env_view(const orc_cfg *c, const orc_bufs *b, int k) {
  orc_env v;
  v.state = b->state + (size_t)k * c->R * NPARAMS * c->C;
  v.leading = b->leading + (size_t)k * c->R;
  v.lastcar = b->lastcar + (size_t)k * c->R;
  v.obs = b->obs + (size_t)k * (2 * c->r + 2 * c->I);
  v.rewards = b->rewards + (size_t)k * c->I;
  v.waiting = b->waiting + (size_t)k * c->r;
  v.passed_dst = b->passed_dst + (size_t)k * c->I;
  return v;
}

void orc_reset_batch(const orc_cfg *c, const orc_bufs *b, const int32_t *phase_init) {
  for (int k = 0; k < c->E; ++k) {
    orc_env v = env_view(c, b, k);
    orc_reset(c, &v, phase_init + (size_t)k * c->I);
  }
}

/* action [E][I]; spawn_off [E+1] into spawn_roads; ticks float32[E]; done u8[E] out.
 * trip buffers optional: trip_times [E][trip_cap], n_trips [E]. */
void orc_step_batch_arch(const orc_cfg *c, const int32_t *dests, const int32_t *phases,
                         const int32_t *nexts, const orc_bufs *b, const int32_t *action,
                         const int64_t *spawn_off, const int32_t *spawn_roads, const float *ticks,
                         float *trip_times, int64_t *n_trips, int64_t trip_cap, int nthreads,
                         int64_t *updates, const float *archetypes, const int32_t *spawn_arch);

void orc_step_batch(const orc_cfg *c, const int32_t *dests, const int32_t *phases,
                    const int32_t *nexts, const orc_bufs *b, const int32_t *action,
                    const int64_t *spawn_off, const int32_t *spawn_roads, const float *ticks,
                    float *trip_times, int64_t *n_trips, int64_t trip_cap, int nthreads,
                    int64_t *updates) {
  orc_step_batch_arch(c, dests, phases, nexts, b, action, spawn_off, spawn_roads, ticks, trip_times, n_trips, trip_cap,
                      nthreads, updates, NULL, NULL);
}

/* archetypes float32[n][10] + spawn_arch int32[] parallel to spawn_roads (both NULL: the config's archetype) */
void orc_step_batch_arch(const orc_cfg *c, const int32_t *dests, const int32_t *phases,
                         const int32_t *nexts, const orc_bufs *b, const int32_t *action,
                         const int64_t *spawn_off, const int32_t *spawn_roads, const float *ticks,
                         float *trip_times, int64_t *n_trips, int64_t trip_cap, int nthreads,
                         int64_t *updates, const float *archetypes, const int32_t *spawn_arch) {
  int64_t total = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1) reduction(+ : total)
#endif
  for (int k = 0; k < c->E; ++k) {
    orc_env v = env_view(c, b, k);
    int64_t u = 0;
    b->done[k] = (uint8_t)orc_step_arch(c, dests, phases, nexts, &v, action + (size_t)k * c->I,
                                        spawn_roads + spawn_off[k], (int)(spawn_off[k + 1] - spawn_off[k]),
                                        ticks[k], trip_times ? trip_times + (size_t)k * trip_cap : NULL,
                                        n_trips ? n_trips + k : NULL, trip_cap, &u, archetypes,
                                        spawn_arch ? spawn_arch + spawn_off[k] : NULL);
    total += u;
  }
  if (updates) *updates += total;
  (void)nthreads;
}

void orc_move_cars_batch(const orc_cfg *c, const int32_t *dests, const int32_t *phases,
                         const int32_t *nexts, const orc_bufs *b) {
  for (int k = 0; k < c->E; ++k) { orc_env v = env_view(c, b, k); orc_move_cars(c, dests, phases, nexts, &v); }
}

void orc_advance_batch(const orc_cfg *c, const int32_t *dests, const int32_t *nexts,
                       const orc_bufs *b, const float *ticks, float *trip_times,
                       int64_t *n_trips, int64_t trip_cap) {
  for (int k = 0; k < c->E; ++k) {
    orc_env v = env_view(c, b, k);
    b->done[k] = (uint8_t)orc_advance(c, dests, nexts, &v, ticks[k],
                                      trip_times ? trip_times + (size_t)k * trip_cap : NULL,
                                      n_trips ? n_trips + k : NULL, trip_cap);
  }
}

void orc_remi_batch(const orc_cfg *c, const int32_t *dests, const int32_t *phases, const orc_bufs *b) {
  for (int k = 0; k < c->E; ++k) { orc_env v = env_view(c, b, k); orc_remi(c, dests, phases, &v); }
}

void orc_cars_on_roads_batch(const orc_cfg *c, const orc_bufs *b, int32_t *out) {
  for (int k = 0; k < c->E; ++k) { orc_env v = env_view(c, b, k); orc_cars_on_roads(c, &v, out + (size_t)k * c->R); }
}

int orc_sizeof_cfg(void) { return (int)sizeof(orc_cfg); }
int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
