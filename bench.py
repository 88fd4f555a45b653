#!/usr/bin/env python3
"""Headline benchmark: vehicle-updates/s of the IDM traffic-env tick on MI355X.

    python bench.py --gpus N --steps K --warmup W
    N > 1 without WORLD_SIZE in the environment: this process touches no GPU; it starts the N ranks
    itself (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...),
    relays rank 0's JSON line and exits with the ranks' return code.  Started under torchrun
    (WORLD_SIZE set) it is one of those ranks.

One "step" = one env tick (TrafficEnv._step, reference traffic_env.py:224-248) over EVERY env of
the batch: the move kernel + the advance kernel.  Workload = BASELINE.json's headline single-GPU
configuration (cfg2: 4096 envs/GPU, 16x16 grid, 64-car roads) on synthetic fixed-spawn traffic
generated on the device (gym_traffic/workload.py); per-GPU work is fixed as N grows (weak scaling),
envs are sharded by id with no collective in the step; for N > 1 a snapshot of (obs, reward, done)
is gathered to rank 0 over RCCL every 10 ticks on a side stream (cfg3).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      dominant kernel (k_move): algorithmic bytes per launch / its mean duration measured
                with HIP events on the launch stream inside the timed region, against 8 TB/s;
  cpu_baseline  the CPU oracle (a port of the reference's algorithm, oracle/idm_oracle.c) timed on
                this host's cores on a bounded sample of the same workload (N = 1, rank 0 only).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "traffic-env_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_STREAM_GBS = 6290.0     # what a streaming copy reaches on this part (same guide: 6.29 TB/s measured, 79 %)
GATHER_EVERY = 10           # ticks between (obs, reward, done) snapshots to rank 0 (one agent step)


def host_threads():
    """Threads this process may really use: the affinity mask, capped by the cgroup CPU quota (the
    GPU boxes expose every core of the host but grant a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_baseline(name, budget_s=12.0, max_ticks=20000):
    """Time the oracle on a bounded sample of the same workload: 8 envs per host thread (at most
    256), same prefill / spawn / light rules, for about `budget_s` seconds."""
    import numpy as np
    from gym_traffic import workload as wl
    from oracle.oracle import OracleEnv
    from gym_traffic.envs.roadgraph import GridRoad

    c = wl.CONFIGS[name]
    threads = host_threads()
    envs = max(1, min(256, 8 * threads, c["envs"]))
    g = GridRoad(c["m"], c["n"], c["length"])
    g.generate_entrypoints(0)
    orc = OracleEnv(c["m"], c["n"], c["length"], c["capacity"], g.dest, g.phases, g.nexts, n_envs=envs)
    orc.reset(np.zeros(orc.I, np.int32))
    x, v, leading, lastcar = wl.prefill_one_env(c["m"], c["n"], c["length"], c["capacity"], c["prefill"], c["gap"])
    w = np.zeros_like(x)
    for k in range(envs):
        orc.load_planes(k, x, v, w, leading, lastcar)
    env_ids = np.arange(envs)
    # the spawn rule has period SPAWN_PERIOD, the light rule 2*LIGHT_PERIOD: build one period of
    # schedules up front so that only the oracle's C code is inside the timed loop
    period = int(np.lcm(wl.SPAWN_PERIOD, 2 * wl.LIGHT_PERIOD))
    sched = []
    for t in range(period):
        roads = np.asarray(wl.spawn_roads_for_tick(g.entrypoints, t), np.int32)
        off = np.arange(envs + 1, dtype=np.int64) * len(roads)
        sched.append((wl.cycle_actions(env_ids, orc.I, t), (off, np.tile(roads, envs))))
    for t in range(3):                                   # warm the code and the caches
        orc.step(*sched[t], nthreads=threads)
    base = orc.vehicle_updates
    t0 = time.perf_counter()
    ticks = 0
    for t in range(3, max_ticks):
        orc.step(*sched[t % period], nthreads=threads)
        ticks += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    updates = orc.vehicle_updates - base
    # the same sample on ONE thread for a few seconds (report both ratios)
    base1 = orc.vehicle_updates
    t1 = time.perf_counter()
    k = 0
    while time.perf_counter() - t1 < 3.0 and k < max_ticks:
        orc.step(*sched[(3 + ticks + k) % period], nthreads=1)
        k += 1
    dt1 = time.perf_counter() - t1
    one = (orc.vehicle_updates - base1) / dt1
    return {"value": updates / dt, "unit": "vehicle-updates/s", "cores": threads, "kind": "port",
            "one_core_value": one,
            "sample": "%d envs x %d ticks of %s (%.3g vehicle-updates in %.1f s), OpenMP over envs "
                      "on %d threads, oracle/idm_oracle.c built gcc -O3 -ffp-contract=off -fopenmp (oracle/Makefile); "
                      "then %d ticks on 1 thread (%.1f s)"
                      % (envs, ticks, name, updates, dt, threads, k, dt1)}


def load_pmc_traffic(name, kernel, field="hbm_bytes_per_tick"):
    """A figure of the kernel that moves the cars from the committed rocprofv3 PMC summary (profiles/pmc_<cfg>.json,
    tools/pmc_summary.py) - only if it was taken for this workload, this kernel AND this kernel's machine code: the
    summary carries the hash of the kernel's instructions in the library that was profiled (taken on the box by
    tools/profile_round.sh, tools/kernel_hash.py); the library this process drives must have the same.  Otherwise null."""
    path = os.path.join(ROOT, "profiles", "pmc_%s.json" % name)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from kernel_hash import kernel_hashes
        with open(path) as f:
            d = json.load(f)
        if d.get("kernel") != kernel or not d.get("kernel_hash") or d.get("kernel_hash") != kernel_hashes().get(kernel):
            return None
        return d.get(field)
    except (OSError, ValueError, ImportError, RuntimeError, subprocess.CalledProcessError):
        return None


VALU_PEAK_GINST = 1024 * 2.4 / 2.0   # wave64 vector instructions per ns the chip can issue: 256 CUs x 4 SIMDs at
#                                      2.4 GHz, one per 2 cycles per SIMD with >= 2 wavefronts resident (the guide's
#                                      `v_fma_f32 (wave64) 2 cyc`) = 1228.8 G wave-instructions / s


def roofline(a, c, eng, E, prof, prof_updates, chunk, dt, updates):
    """The dominant kernel against the roofline that bounds it.

    HBM-bound kernels (the streaming move kernels): ALGORITHMIC bytes of one launch / its mean duration / 8 TB/s.
    One launch that takes the cars through T ticks (k_move_tt: T = 2) reads and writes every live car ONCE -
    16 B per car - and the per-road words once per tick - 48 B per road and tick (SURVEY.md 8d's figures):
        bytes per launch = 16 N_live + T 48 E R.
    (SURVEY's per-tick model, 16 B per vehicle-UPDATE, would count the cars T times; a T-tick pass is the
    point of moving them once, so that figure is kept as `per_tick_model` and is not a fraction of anything.)
    `traffic` is the PMC measurement of the same launch (profiles/pmc_<cfg>.json, only while the kernel sources are
    the profiled ones); `frac` must agree with traffic / launch time / peak within a few percent.

    k_res keeps the cars in LDS for all ticks of a call and touches HBM twice per call: it is bound by vector
    instruction issue, so its roofline is wave-instructions per second against the chip's issue rate (the
    instruction count per tick comes from the PMC pass, SQ_INSTS_VALU)."""
    from gym_traffic import workload as wl
    K = prof["ticks"]
    live = prof_updates / max(1, K)                                     # mean live cars per tick (roofline pass)
    move_ms = prof["move_ms"] / max(1, K)                               # per tick
    rest_ms = prof["advance_ms"] / max(1, K)
    kernel = eng.step_kernel()
    if 2 * (eng.pair_ticks()) >= K and kernel.startswith("k_move_tt"):
        kernel = "k_move_tts" if kernel == "k_move_tts" else "k_move_tt"      # (tts: two wavefronts per tile, mid-size launches)
    call = min(chunk, K)                                                # ticks per tfx_step call
    tpl = (call if kernel == "k_res" else 2 if kernel in ("k_move_tt", "k_move_tts") else 1)     # ticks per LAUNCH
    passes = 1                                                                  # trips of the cars through HBM per launch
    launch_ms = move_ms * tpl
    full = E == c["envs"]
    if kernel == "k_res":
        insts = load_pmc_traffic(a.config, kernel, "valu_insts_per_tick") if full else None
        ach = insts / (move_ms * 1e6) if insts and move_ms > 0 else None     # G wave-instructions / s
        return {"bound": "valu", "kernel": kernel, "achieved": ach, "peak": VALU_PEAK_GINST,
                "unit": "G wave64-instructions/s", "frac": ach / VALU_PEAK_GINST if ach else None,
                "traffic": None, "valu_instructions_per_tick": insts, "launch_ms": launch_ms,
                "ticks_per_launch": tpl, "ticks_timed": K,
                "note": "LDS-resident multi-tick kernel: HBM is touched at the start and the end of a call only; "
                        "the bound is vector-instruction issue (count from the PMC pass x 2 cycles per wave64 "
                        "instruction / 1024 SIMDs / 2.4 GHz)"}
    road_ticks = E * eng.R
    alg = 16.0 * live * passes + tpl * 48.0 * road_ticks
    achieved = alg / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    per_tick = load_pmc_traffic(a.config, kernel, "hbm_bytes_per_tick") if full else None
    traffic = per_tick * tpl if per_tick else None
    per_tick_model = tpl * (16.0 * live + 48.0 * road_ticks)
    region_bytes = 16.0 * (updates / max(1, a.steps)) * passes + tpl * (48.0 * road_ticks + 32.0 * E * eng.I)
    region_s = tpl * dt / max(1, a.steps)
    return {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_frac": traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic and launch_ms > 0 else None,
            "frac_of_measured_stream_peak": achieved / HBM_STREAM_GBS,
            "algorithmic_bytes_per_launch": alg, "launch_ms": launch_ms, "ticks_per_launch": tpl,
            "passes_per_launch": passes, "ticks_timed": K, "rest_of_tick_ms": rest_ms,
            "per_tick_model": {"bytes_per_launch": per_tick_model,
                               "rate_GBs": per_tick_model / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0,
                               "note": "SURVEY 8d's 16 B per vehicle-UPDATE x the updates of the launch: counts the "
                                       "cars once per tick although a two-tick pass moves them once; not a fraction "
                                       "of the HBM peak"},
            # the whole timed region against the same peak, same byte model: per pass every live car once (16 B), the
            # per-road / per-intersection words once per tick, over the wall time of the region (median repeat)
            "timed_region": {"GBs": region_bytes / region_s / 1e9, "frac": region_bytes / region_s / 1e9 / HBM_PEAK_GBS},
            "measured": "HIP events on the launch stream around every launch of the kernel, in a second pass over "
                        "the same K ticks right after the timed regions (no events inside a timed region; the env "
                        "range is not split over two streams while a launch is timed)"}


def numpy_baseline(name, budget_s=8.0):
    """SURVEY 8d's second CPU leg: the same tick as batched NumPy array arithmetic on ONE core (oracle/numpy_env.py)."""
    from oracle.numpy_env import time_config
    return time_config(name, budget_s)


class stdout_to_stderr(object):
    """File descriptor 1 -> stderr for the duration (native libraries write to the descriptor, not to sys.stdout)."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(a, argv):
    """--gpus N > 1 and no WORLD_SIZE: start the N ranks as a child job, print rank 0's line, return
    the job's exit code.  Nothing in this process initialises a GPU (torch.cuda.device_count() does
    not on this image), nothing is exec'ed, a failed job is not retried."""
    import subprocess
    import torch
    rehearsal = os.environ.get("TFX_BENCH_REHEARSAL") == "1"
    have = torch.cuda.device_count()
    if not rehearsal and not a.selftest_launcher and have < a.gpus:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible (TFX_BENCH_REHEARSAL=1 runs every "
                         "rank on GPU 0 over gloo as a dry run)\n" % (a.gpus, have))
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    job = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, universal_newlines=True)
    line = None
    for out in job.stdout:
        if out.startswith('{"metric"'):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = job.wait()
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks finished without a result line\n")
        return 3
    if line is not None and rc == 0:
        print(line, flush=True)
    return rc


def selftest_rank(a):
    """One rank of `--selftest-launcher`: gloo on the CPU, synthetic snapshots through RolloutGather,
    the bench's reductions and the single JSON line - the N > 1 path minus the GPU."""
    import torch
    import torch.distributed as dist
    from gym_traffic.distributed import RolloutGather
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if rank == a.selftest_fail_rank:
        return 7
    if world > 1:
        dist.init_process_group("gloo")
    n, L, I = 3, 5, 2
    g = RolloutGather(n, L, I, "cpu")
    ok = True
    for step in range(4):
        base = 1000 * rank + 10 * step
        g.start(torch.full((n, L), base, dtype=torch.int32), torch.full((n, I), base + 0.5),
                torch.full((n,), step % 2, dtype=torch.uint8))
    res = g.result()
    if rank == 0:
        obs, rew, done = res
        want = torch.tensor([1000 * k + 30 for k in range(world) for _ in range(n)], dtype=torch.int32)
        ok = (obs.shape == (world * n, L) and bool((obs[:, 0] == want).all()) and
              bool((rew[:, 1] == want.float() + 0.5).all()) and bool((done == 1).all()) and
              g.collectives == (4 if world > 1 else 0))
    tt = torch.tensor([1.0 + rank], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "launcher_selftest", "value": float(ok), "unit": "ok", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "t_max": float(tt.item()),
                          "data": "none: plumbing check, no kernel ran"}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--stream-priority", type=int, default=None,
                    help="run the ticks on a torch stream of this priority (-1 = high; the default with a process "
                         "group) instead of the default stream")
    ap.add_argument("--rccl-one-rank", action="store_true",
                    help="N = 1 with everything a rank of an N > 1 run does: an RCCL process group (of one rank), the "
                         "snapshot + gather every %d ticks, barriers and reductions" % GATHER_EVERY)
    ap.add_argument("--no-numpy-baseline", action="store_true")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of --steps ticks each (fenced); the line reports their median and the spread")
    ap.add_argument("--no-agent-steps", action="store_true",
                    help="skip the measurement of fused agent decisions (profile runs: only the timed ticks' launches)")
    ap.add_argument("--settle", type=int, default=None,
                    help="untimed ticks run right after the prefill, as part of the workload's setup and before the W "
                         "warm-up steps (default: the workload's, gym_traffic/workload.py SETTLE_TICKS; 0 = none)")
    ap.add_argument("--call-ticks", type=int, default=0,
                    help="ticks per tfx_step call in the timed region (default: all K in one call; with the N > 1 "
                         "gather: %d, one agent step)" % GATHER_EVERY)
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU-only check of the N > 1 plumbing (parent launch, rendezvous, gather, relay); "
                         "runs no kernel and measures nothing")
    ap.add_argument("--selftest-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a, sys.argv[1:]))
    if a.selftest_launcher:
        sys.exit(selftest_rank(a))

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist
    from gym_traffic import workload as wl
    from gym_traffic.distributed import RolloutGather

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit("--gpus %d needs WORLD_SIZE=%d (launch with torch.distributed.run)" % (a.gpus, a.gpus))
    # TFX_BENCH_REHEARSAL=1: every rank on GPU 0 with gloo - a one-GPU dry run of the N > 1 path
    rehearsal = os.environ.get("TFX_BENCH_REHEARSAL") == "1"
    local = 0 if rehearsal else local
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist_on = world > 1 or a.rccl_one_rank
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a version banner on STDOUT when its communicator comes up (gloo a line per rank): stdout carries
        # the one JSON line and nothing else, so file descriptor 1 points at stderr until the communicator exists
        with stdout_to_stderr():
            if rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=device)
            dist.barrier()

    # With a process group the ticks run on a HIGH-priority stream (the handle's second stream is then a low-priority
    # one): HIP shares its hardware queues among the streams of one priority level, and RCCL's own streams and
    # torch's pool are of normal priority - a collective kernel that waits for its peers must never sit in the
    # hardware queue the car passes go through.
    prio = a.stream_priority if a.stream_priority is not None else (-1 if dist_on else None)
    if prio is not None:
        torch.cuda.set_stream(torch.cuda.Stream(device, priority=prio))
    c = wl.CONFIGS[a.config]
    E = int(a.envs if a.envs is not None else c["envs"])
    eng = wl.setup_engine(a.config, device=device, envs=E, env_id_offset=rank * E)
    # The prefill puts the SAME platoon on every road of every env, and all of them brake at once: for the first ~25
    # ticks every car of the batch is accelerating or braking, and the chip runs those ticks at a lower clock (same
    # instructions, same bytes, same wave-cycles per launch - DESIGN.md 5).  The workload therefore includes settle
    # ticks: the timed region measures traffic that has found its queues, whatever warm-up the caller asks for.
    settle = wl.SETTLE_TICKS.get(a.config, 0) if a.settle is None else a.settle
    for _ in range(settle // 50):
        eng.step(50)
    if settle % 50:
        eng.step(settle % 50)
    torch.cuda.synchronize(device)
    gather = None
    if dist_on and not a.no_gather:
        gather = RolloutGather(E, eng.obs_len, eng.I, device, single_rank_collective=a.rccl_one_rank)

    chunk = GATHER_EVERY if gather is not None else (a.call_ticks if a.call_ticks > 0 else 1 << 30)

    def run(n):
        done = 0
        while done < n:
            k = min(chunk, n - done)
            eng.step(k)
            done += k
            if gather is not None and done % GATHER_EVERY == 0:
                gather.start(eng.obs, eng.rewards, eng.done)
        if gather is not None:
            gather.wait()

    def fence():
        torch.cuda.synchronize(device)
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize(device)

    # Timed regions: nothing but the K ticks (no HIP events in them), REPEATS times, each fenced on both sides (barrier +
    # device synchronisation); `value` is the MEDIAN region, `spread` the slowest and the fastest - one region of 20 ticks
    # is 8 ms, and box to box and run to run the same binary moves by a few percent.
    run(a.warmup)

    pair_ticks0, split_ticks0 = eng.pair_ticks(), eng.split_ticks()
    regions = []
    for _ in range(max(1, a.repeats)):
        eng.reset_counters()
        fence()
        t0 = time.perf_counter()
        run(a.steps)
        fence()
        regions.append((time.perf_counter() - t0, eng.vehicle_updates()))
    pair_ticks = (eng.pair_ticks() - pair_ticks0) // len(regions)
    split_ticks = (eng.split_ticks() - split_ticks0) // len(regions)

    # Roofline pass: the SAME K ticks again, now with HIP events on the launch stream around every launch of the
    # kernel that moves the cars (tfx_profile).  While it is timed a launch owns the chip: the handle does not
    # split the env range over two streams in this pass (tfx_split_ticks), so the duration is the kernel's own.
    eng.profile(a.steps)
    eng.reset_counters()
    fence()
    run(a.steps)
    fence()
    prof = eng.profile_read()
    prof_updates = eng.vehicle_updates()
    eng.profile(0)

    # Agent decisions, MEASURED: the fused 10-tick decision (Repeater + Remi, tfx_agent_step) under the same rules, on
    # a second handle brought to the state the first timed region started from (prefill, settle ticks, warm-up): late in a
    # long run the benchmark's entry roads run full, and an env that overflows in the first tick of a decision stands
    # still for the rest of it (`if done: break`) - which would time decisions that do nothing
    n_dec, dt_agent, agent_done = 0, 0.0, 0
    if not a.no_agent_steps:
        eng2 = wl.setup_engine(a.config, device=device, envs=E, env_id_offset=rank * E)
        for n in [50] * (settle // 50) + [settle % 50, a.warmup]:
            if n:
                eng2.step(n)
        n_dec = 5
        eng2.agent_step(GATHER_EVERY, remi=True)
        fence()
        t0 = time.perf_counter()
        for _ in range(n_dec):
            adone = eng2.agent_step(GATHER_EVERY, remi=True)[2]
        fence()
        dt_agent = time.perf_counter() - t0
        agent_done = int(adone.sum().item())
        del eng2

    red_dev = torch.device("cpu") if rehearsal else device
    tt = torch.tensor([r[0] for r in regions] + [dt_agent], dtype=torch.float64, device=red_dev)
    uu = torch.tensor([float(r[1]) for r in regions], dtype=torch.float64, device=red_dev)
    t_min = tt.clone()
    if dist_on:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)        # every region: the slowest rank's time
        dist.all_reduce(t_min, op=dist.ReduceOp.MIN)
        dist.all_reduce(uu, op=dist.ReduceOp.SUM)
    times, totals = tt[:-1].tolist(), uu.tolist()
    rates = [u / t for u, t in zip(totals, times)]
    order = sorted(range(len(rates)), key=lambda i: rates[i])
    mid = order[(len(order) - 1) // 2]                   # the median region (the lower one of an even count)
    dt_max, total_updates = times[mid], totals[mid]
    dt, updates = regions[mid]
    dt_agent_max = float(tt[-1].item())

    if rank == 0:
        K = a.steps
        out = {
            "metric": "vehicle_updates_per_sec",
            "value": total_updates / dt_max,
            "unit": "vehicle-updates/s",
            "n_gpus": world,
            "steps": K,
            "warmup": a.warmup,
            "ms_per_step": dt_max / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wl.describe(a.config), "settle_ticks": settle,
                       "envs_per_gpu": E,
                       "grid": "%dx%d" % (c["m"], c["n"]), "cars_per_road_max": c["capacity"] - 2,
                       "car_layout": eng.layout,
                       "parallelism": "env-sharded x%d%s" % (
                           world, ", one %s gather of (obs|reward|done) to rank 0 every %d ticks, double-"
                           "buffered on a side stream" % ("gloo (rehearsal)" if rehearsal else "RCCL", GATHER_EVERY)
                           if gather is not None else "")},
            # every timed region of K ticks (fenced), the line's value being their median
            "repeats": len(rates), "spread": [min(rates), max(rates)],
            "ms_per_step_spread": [min(times) / K * 1e3, max(times) / K * 1e3],
            "env_steps_per_sec": world * E * K / dt_max,
            # measured through tfx_agent_step (one decision = %d ticks + remi), not derived from the ticks above
            "agent_steps_per_sec": world * E * n_dec / dt_agent_max if n_dec else None,
            "agent_decision_ms": dt_agent_max / n_dec * 1e3 if n_dec else None,
            "agent_decisions_timed": n_dec, "agent_envs_done_in_last_decision": agent_done,
            "regions": [{"ms": t * 1e3, "vehicle_updates": u} for t, u in zip(times, totals)],
            "mean_live_cars_per_road": updates / K / (E * eng.R),
            "ticks_in_two_tick_passes": pair_ticks,
            "ticks_split_over_two_streams": split_ticks,
            "ticks_per_call": min(chunk, K),
            "roofline": roofline(a, c, eng, E, prof, prof_updates, chunk, dt_max, updates),
        }
        if dist_on:
            # what a scaling curve needs beside the aggregate: the ranks' own times for the median region
            out["per_rank_ms_per_step"] = {"max": dt_max / K * 1e3, "min": float(t_min[mid].item()) / K * 1e3}
            out["t_max_over_t_min"] = dt_max / max(1e-12, float(t_min[mid].item()))
            out["rccl_ranks_seen"] = dist.get_world_size() if not rehearsal else 0
            out["gather_ms_per_snapshot"] = gather.ms_per_snapshot() if gather is not None else None
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.config)
            if not a.no_numpy_baseline:
                out["cpu_baseline"]["numpy_batched_one_core"] = numpy_baseline(a.config)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
