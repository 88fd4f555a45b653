#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh <outdir under gpurun_out>
# kernel-trace --stats and the PMC passes (FETCH_SIZE and WRITE_SIZE in passes of their own, as the
# guide prescribes; SQ counters in a third) for cfg2, cfg1, cfg4, plus the un-profiled bench lines.
# tools/pmc_summary.py condenses the directories into profiles/.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-prof}
mkdir -p $O
# the hash of every kernel's machine code in the library that runs below: travels with the counters (pmc_summary.py
# stamps its summaries with it and refuses counters that come without)
python3 $R/tools/kernel_hash.py > $O/kernel_hashes.json || { echo "kernel_hash failed"; exit 1; }
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
prof() {  # name, rocprof args..., -- bench args
  local name=$1; shift
  local rargs=(); while [ "$1" != "--" ]; do rargs+=("$1"); shift; done; shift
  rocprofv3 "${rargs[@]}" -d $O/$name -o p --output-format csv -- python3 $R/bench.py --steps 50 --warmup 10 --repeats 1 --no-agent-steps --no-cpu-baseline "$@" > $O/$name.log 2>&1 || echo "FAILED $name"
  echo "done $name"
}
# BENCH_ONLY=1: only the un-profiled bench lines below
[ "${BENCH_ONLY:-0}" = 1 ] || {
# Per-kernel numbers are taken with the env range in ONE piece (TFX_SPLIT=0): a launch that shares the chip with the
# other half's launches has no duration of its own.  cfg2_split_kt is the default (split) run for the record.
export TFX_SPLIT=0
for cfg in cfg2 cfg1 cfg4; do
  prof ${cfg}_kt --kernel-trace --stats -- --config $cfg
  prof ${cfg}_fetch --kernel-trace --pmc FETCH_SIZE -- --config $cfg
  prof ${cfg}_write --kernel-trace --pmc WRITE_SIZE -- --config $cfg
  prof ${cfg}_sq --kernel-trace --pmc $SQ -- --config $cfg
done
# the tick-by-tick kernels next to the two-tick passes (k_move_t's own numbers)
TFX_PAIRS=0 prof cfg2_pertick_kt --kernel-trace --stats -- --config cfg2
TFX_PAIRS=0 prof cfg2_pertick_fetch --kernel-trace --pmc FETCH_SIZE -- --config cfg2
TFX_PAIRS=0 prof cfg2_pertick_write --kernel-trace --pmc WRITE_SIZE -- --config cfg2
unset TFX_SPLIT
prof cfg2_split_kt --kernel-trace --stats -- --config cfg2
}
cd $R
# (cfg0 / cfg1: k_res runs the whole timed region as ONE launch of 2 ms per 200 ticks: timed over 2000)
for cfg in cfg2 cfg1 cfg0 cfg4; do
  WU=20; ST=200; case $cfg in cfg0|cfg1) WU=600; ST=2000;; esac
  python3 bench.py --config $cfg --steps $ST --warmup $WU $( [ $cfg = cfg2 ] || echo --no-cpu-baseline ) > $O/bench_$cfg.json 2> $O/bench_$cfg.err
  echo "bench $cfg: $(python3 -c "import json;d=json.load(open('$O/bench_$cfg.json'));print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])")"
done
TFX_PAIRS=0 python3 bench.py --config cfg2 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_cfg2_pertick.json 2>/dev/null
TFX_PAIRS=0 python3 bench.py --config cfg4 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_cfg4_pertick.json 2>/dev/null
TFX_RESIDENT=0 python3 bench.py --config cfg1 --steps 200 --warmup 600 --no-cpu-baseline > $O/bench_cfg1_pertick.json 2>/dev/null
TFX_SPLIT=0 python3 bench.py --config cfg2 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_cfg2_nosplit.json 2>/dev/null
TFX_SPLIT=0 TFX_TAIL=0 python3 bench.py --config cfg2 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_cfg2_nosplit_notail.json 2>/dev/null
python3 bench.py --config cfg1 --envs 4096 --steps 1000 --warmup 600 --no-cpu-baseline > $O/bench_cfg1_4096.json 2>/dev/null
python3 bench.py --config cfg2 --envs 256 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_cfg2_256.json 2>/dev/null
python3 bench.py --config cfg2 --envs 32768 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_cfg2_32768.json 2>/dev/null
( python3 tools/bench_resident.py; python3 tools/bench_single_env.py ) 2>&1 | grep -v amdgpu.ids > $O/small_configs.txt
( python3 tools/run_cfg4.py; python3 tools/c4_loop.py 1; python3 tools/c4_loop.py 16 ) 2>&1 | grep -v amdgpu.ids > $O/cfg4_closed_loop.txt
( python3 tools/bench_agent_step.py cfg2; TFX_PAIRS=0 python3 tools/bench_agent_step.py cfg2 ) 2>&1 | grep -v amdgpu.ids > $O/agent_step_cfg2.txt
# the driver's short run, with and without the workload's settle ticks; one rank with everything a rank of an N > 1 run does
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg2_20steps.json 2>/dev/null
python3 bench.py --steps 20 --warmup 5 --settle 0 --no-cpu-baseline > $O/bench_cfg2_20steps_nosettle.json 2>/dev/null
python3 bench.py --steps 200 --warmup 20 --settle 0 --no-cpu-baseline > $O/bench_cfg2_nosettle.json 2>/dev/null
python3 bench.py --rccl-one-rank --no-cpu-baseline > $O/bench_cfg2_rccl_one_rank.json 2>/dev/null
( python3 tools/bench_validate.py; TFX_PAIRS=0 python3 tools/bench_validate.py; python3 tools/bench_archetypes.py ) 2>&1 | grep -v amdgpu.ids > $O/validate_and_archetypes_cfg2.txt
echo finished
