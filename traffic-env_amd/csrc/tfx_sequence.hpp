// tfx_sequence.hpp - the launch sequences of tfx_step and tfx_agent_step: pairs of ticks, the env range in two halves on
// two streams, the agent step (host side; included by tfx_hip.hip).
#pragma once
#include "tfx_launch.hpp"

namespace {

// the second stream of a split call and the events that fork it from / join it to the caller's stream.
// The second stream must not share a hardware queue with the caller's: HIP hands its hardware queues (4 per priority
// level by default) to streams as they are created and shares them from then on, and two streams on one queue run
// their kernels in order - no overlap.  Measured at cfg2 in a process that had initialised RCCL before the handle's
// first split call (as every rank of `bench.py --gpus N` does): a plain second stream 4.43-4.47e11 vehicle-updates/s
// against 5.1-5.2e11 without RCCL; a second stream of ANOTHER priority level - its own pool of queues - 5.10-5.11e11
// either way (low priority: 4.91e11; a CU-masked stream: 4.43e11 either way).
int ensure_split(tfx_handle h, hipStream_t caller) {
  int lo = 0, hi = 0, cp = 0;
  HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));  // (numerically: hi <= 0 <= lo)
  if (hipStreamGetPriority(caller, &cp) != hipSuccess) cp = 0;
  const int want = (cp == hi) ? lo : hi;   // high priority; low if high is the caller's own level
  if (h->split_stream && h->split_prio == want) return TFX_OK;
  if (h->split_stream) {
    HIPCHK(hipStreamSynchronize(h->split_stream));
    HIPCHK(hipStreamDestroy(h->split_stream));
    h->split_stream = nullptr;
  }
  HIPCHK(hipStreamCreateWithPriority(&h->split_stream, hipStreamNonBlocking, want));
  h->split_prio = want;
  if (h->split_fork) return TFX_OK;
  HIPCHK(hipEventCreateWithFlags(&h->split_fork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&h->split_join, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&h->split_stagger, hipEventDisableTiming));
  return TFX_OK;
}

// Whatever a launch sequence changes in the handle while it enqueues - the device block (agent mode, reward
// accumulation, the sub-range of one half), the half being enqueued - is put back on EVERY exit, and a second stream
// that was forked is joined back into the caller's stream, so that work already enqueued there stays ordered before
// whatever the caller enqueues next: an error in the middle of a sequence leaves a handle the next call can use.
struct SeqGuard {
  tfx_handle h;
  Dev keep;
  hipStream_t user = nullptr;
  bool forked = false;
  explicit SeqGuard(tfx_handle hh) : h(hh), keep(hh->d) {}
  int fork(hipStream_t st) {
    user = st;
    HIPCHK(hipEventRecord(h->split_fork, st));
    HIPCHK(hipStreamWaitEvent(h->split_stream, h->split_fork, 0));
    forked = true;
    return TFX_OK;
  }
  int join() {
    if (!forked) return TFX_OK;
    forked = false;
    HIPCHK(hipEventRecord(h->split_join, h->split_stream));
    HIPCHK(hipStreamWaitEvent(user, h->split_join, 0));
    return TFX_OK;
  }
  ~SeqGuard() {
    h->d = keep;
    h->split_half = -1;
    h->split_first = false;
    h->size_only = false;
    if (forked) {  // (an error exit: best effort, the error being reported is the first one)
      const std::string first = g_err;
      (void)join();
      g_err = first;
    }
  }
};

// the launches of one agent step, in order, on `st`
// split: the ticks run as two halves of the env range, the second on the handle's own stream (as step_chunk does for
// tfx_step; launched eagerly - a batch big enough to split is not bound by its launches)
int agent_sequence(tfx_handle h, int n_ticks, int remi, float *aobs, float *areward, uint8_t *adone,
                   hipStream_t st, long long &n_fused, long long &n_pair, bool split = false) {
  Dev &d = h->d;
  n_fused = n_pair = 0;
  SeqGuard guard(h);
  if (res_usable(h, n_ticks)) {
    // every tick of the decision AND its tail (remi, observation, rewards, done flags) in one launch
    d.agent_mode = 1;
    d.accum_rewards = remi ? 0 : 1;
    const int rc = launch_res(h, n_ticks, st, 1, remi, aobs, areward, adone);
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
    // (the second half's clock is copied on the CALLER's stream, ahead of the fork: copied on the second stream it
    // raced with the first half's kernels, which move the clock on - a new stream's first launch can take longer to
    // start than a small batch's whole pair)
    hipLaunchKernelGGL(k_clock_copy, dim3(1), dim3(1), 0, st, whole.tickA, whole.tickB, h->tick2);
    HIPCHK(hipGetLastError());
    if (int frc = guard.fork(st)) return frc;
  }
  hipStream_t user_st = st;
  // (the halves' launches are submitted pair by pair, alternately: see step_chunk)
  Dev halves[2] = {whole, whole};
  if (split) {
    const int n0 = whole.E / 2;
    halves[0] = sub_dev(h, 0, n0, nullptr);
    halves[1] = sub_dev(h, n0, whole.E - n0, h->tick2);
  }
  for (int t0 = 0; t0 < n_ticks && rc == TFX_OK; t0 += 2) {
    for (int half = 0; half < (split ? 2 : 1) && rc == TFX_OK; ++half) {
      if (split) {
        h->d = halves[half];
        st = half == 0 ? user_st : h->split_stream;
        h->split_half = half;
        h->split_first = t0 == 0;
      }
      const bool tt = pairs_usable(h);
      int t = t0;
      if (tt && t + 1 < n_ticks) {
        // a two-tick pass (tfx_move_tt.hpp); envs in which the first tick of a pair could overflow take the pair one
        // tick at a time (k_risk)
        rc = launch_inputs(h, st);
        // (k_tail evaluates the bound for the pair that follows it: only the first pair pays a launch of its own)
        const bool tail = tail_usable(h);
        if (rc == TFX_OK && !(tail && t > 0)) rc = launch_risk(h, t, st);
        if (rc == TFX_OK) rc = launch_move_tt<true, true>(h, t, st, 0, tail);
        if (rc == TFX_OK && tail) {
          // the rest of the pair in one launch (the envs k_risk sorted out get their second tick inside it)
          rc = launch_tail(h, t, st, true, (t + 3 < n_ticks ? TAIL_RISK_NEXT : 0) | (t + 2 >= n_ticks ? TAIL_LAST : 0));
        } else {
          if (rc == TFX_OK) rc = launch_advance(h, t, st);
          if (rc == TFX_OK) rc = launch_inputs(h, st);
          if (rc == TFX_OK) rc = launch_edge<true>(h, t + 1, st);
          if (rc == TFX_OK) rc = launch_move_tt<false, true>(h, t + 1, st, 1);
          if (rc == TFX_OK) rc = launch_advance(h, t + 1, st);
        }
        if (rc == TFX_OK) n_pair += 2;
        t += 2;
      }
      for (; t < n_ticks && t < t0 + 2 && rc == TFX_OK; ++t) {
        rc = launch_inputs(h, st);
        if (rc == TFX_OK) rc = (tt && !single_tick_ts(h)) ? launch_move_tt<false, true>(h, t, st) : launch_move(h, t, st);
        if (rc == TFX_OK) rc = launch_advance(h, t, st);
      }
    }
  }
  if (split) h->d = whole;
  h->split_half = -1;
  st = user_st;
  if (rc != TFX_OK) return rc;
  if (split) {
    n_pair /= 2;  // (both halves counted them)
    if (int jrc = guard.join()) return jrc;
  }
  d.agent_mode = guard.keep.agent_mode;  // the tail kernels below run outside the step's tick loop
  d.accum_rewards = guard.keep.accum_rewards;
  if (remi || aobs || areward || adone) {
    hipLaunchKernelGGL(k_agent_tail, dim3(grid_for((long)d.E * (2 * d.r + d.I), h->n_cu)), dim3(256), 0, st, d, remi, aobs,
                       areward, adone, d.agent_first);
    HIPCHK(hipGetLastError());
  }
  return TFX_OK;
}

}  // namespace

namespace {
// the per-tick kernels for ticks [t_lo, t_hi) of a call of n_ticks ticks (t_lo even; default: all of them) of the envs
// h->d describes (the whole handle, or one half of it), on st
int step_range(tfx_handle h, int n_ticks, hipStream_t st, int t_lo = 0, int t_hi = -1) {
  if (t_hi < 0 || t_hi > n_ticks) t_hi = n_ticks;
  int t = t_lo;
  const bool tt = pairs_usable(h);
  if (tt) {
    for (; t + 1 < t_hi; t += 2) {
      const bool timed = h->prof && h->ev_used < h->ev_ticks;
      hipEvent_t *e = timed ? &h->ev[(size_t)h->ev_used * 3] : nullptr;
      if (int rc = launch_inputs(h, st)) return rc;
      if (timed) HIPCHK(hipEventRecord(e[0], st));
      // (from a call's second pair on the pass reads the road state words the k_tail before it left; the last k_tail
      // of the call stores leading / lastcar / hb themselves)
      if (int rc = launch_move_tt<true>(h, t, st, 0, tail_usable(h), t > 0)) return rc;
      if (timed) HIPCHK(hipEventRecord(e[1], st));
      if (tail_usable(h)) {
        if (int rc = launch_tail(h, t, st, false, (t + 2 >= n_ticks ? TAIL_LAST : 0) | (t + 3 >= n_ticks ? TAIL_SYNC : 0))) return rc;
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
  for (; t < t_hi; ++t) {
    const bool timed = h->prof && h->ev_used < h->ev_ticks;
    hipEvent_t *e = timed ? &h->ev[(size_t)h->ev_used * 3] : nullptr;
    if (int rc = launch_inputs(h, st)) return rc;
    if (timed) HIPCHK(hipEventRecord(e[0], st));
    if (int rc = (tt && !single_tick_ts(h)) ? launch_move_tt<false>(h, t, st) : launch_move(h, t, st)) return rc;
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

// n_ticks ticks on the per-tick kernels, the env range in two halves on two streams where that pays
int step_chunk(tfx_handle h, int n_ticks, hipStream_t st) {
  if (split_usable(h, n_ticks)) {
    // fork: the handle's own stream takes the second half of the envs, the caller's stream the first
    if (int rc = ensure_split(h, st)) return rc;
    if (h->grid_tt[1] == 0 || h->grid_tt[0] == 0 || h->grid_tail == 0) {  // grids are sized for the whole range
      h->size_only = true;
      (void)launch_move_tt<true>(h, 0, nullptr);
      (void)launch_move_tt<false>(h, 0, nullptr);
      (void)launch_tail(h, 0, nullptr);
      (void)launch_advance(h, 0, nullptr);
      h->size_only = false;
    }
    SeqGuard guard(h);
    const Dev whole = h->d;
    const int n0 = whole.E / 2;
    const long long pair0 = h->pair_ticks, tail0 = h->tail_ticks;
    // (the clock copy runs on the caller's stream, ahead of the fork: see agent_sequence)
    hipLaunchKernelGGL(k_clock_copy, dim3(1), dim3(1), 0, st, whole.tickA, whole.tickB, h->tick2);
    HIPCHK(hipGetLastError());
    if (int frc = guard.fork(st)) return frc;
    int rc = TFX_OK;
    // The two halves' launches are submitted pair by pair, alternately.  (Submitted half after half - all of the first
    // half's launches, then the second's - the second stream's first launch reached the device only after the host had
    // queued the whole first half: under rocprofv3 the first half ran FIVE pairs alone at the start of a 50-tick call
    // and the second five alone at its end, each at 0.47 ms per half pair instead of the 0.39 of two halves side by
    // side; profiles/r04_split_submission_order.txt.)
    const Dev halves[2] = {sub_dev(h, 0, n0, nullptr), sub_dev(h, n0, whole.E - n0, h->tick2)};
    for (int t = 0; t < n_ticks && rc == TFX_OK; t += 2) {
      for (int half = 0; half < 2 && rc == TFX_OK; ++half) {
        h->d = halves[half];
        h->split_half = half;
        h->split_first = t == 0;
        rc = step_range(h, n_ticks, half == 0 ? st : h->split_stream, t, t + 2);
      }
    }
    h->d = whole;
    h->split_half = -1;
    h->pair_ticks = pair0 + (h->pair_ticks - pair0) / 2;  // (both halves counted them)
    h->tail_ticks = tail0 + (h->tail_ticks - tail0) / 2;
    if (rc != TFX_OK) return rc;  // (the guard joins the second stream)
    h->split_ticks += n_ticks;
    return guard.join();
  }
  return step_range(h, n_ticks, st);
}
