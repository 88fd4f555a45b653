#!/bin/bash
# Condenses what tools/profile_round.sh left under gpurun_out/<dir> into profiles/ (run here, after the gpurun call):
#   bash tools/profiles_condense.sh <dir under gpurun_out> [round tag, default r04] [dir of the bench lines, default the same]
# (bench lines taken in a second call - BENCH_ONLY=1, after the PMC summaries exist - carry roofline.traffic)
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/${1:?outdir}; RD=${2:-r04}; B=gpurun_out/${3:-$1}
N="TFX_SPLIT=0 bench.py --config CFG --steps 50 --warmup 10 (after the workload's settle ticks) under rocprofv3 (separate FETCH_SIZE / WRITE_SIZE / SQ passes)"
python3 tools/pmc_summary.py --round $RD --config cfg2 --kt $O/cfg2_kt --pmc $O/cfg2_fetch $O/cfg2_write $O/cfg2_sq --kernel k_move_tt \
  --ticks-per-launch 2 --read-bytes-expected 1.89e9 --note "${N/CFG/cfg2}; k_move_tt<true> = one two-tick pass of all 4096 envs"
python3 tools/pmc_summary.py --round $RD --config cfg4 --kt $O/cfg4_kt --pmc $O/cfg4_fetch $O/cfg4_write $O/cfg4_sq --kernel k_move_tts \
  --ticks-per-launch 2 --note "${N/CFG/cfg4}"
python3 tools/pmc_summary.py --round $RD --config cfg1 --kt $O/cfg1_kt --pmc $O/cfg1_fetch $O/cfg1_write $O/cfg1_sq --kernel k_res \
  --ticks-per-launch 50 --note "${N/CFG/cfg1}; one k_res launch = the 50 timed ticks"
python3 tools/pmc_summary.py --round $RD --config cfg2_pertick --kt $O/cfg2_pertick_kt --pmc $O/cfg2_pertick_fetch $O/cfg2_pertick_write \
  --kernel k_move_t --note "TFX_PAIRS=0 TFX_SPLIT=0 bench.py --config cfg2 --steps 50 --warmup 10 under rocprofv3"
python3 tools/pmc_summary.py --round $RD --config cfg2_split --kt $O/cfg2_split_kt --note "the default (split) run under rocprofv3 --kernel-trace --stats"
for f in $B/bench_*.json; do b=$(basename $f); [ -s $f ] && cp $f profiles/${RD}_$b; done
for t in small_configs cfg4_closed_loop agent_step_cfg2 validate_and_archetypes_cfg2; do [ -s $B/$t.txt ] && cp $B/$t.txt profiles/${RD}_$t.txt; done
cp $O/kernel_hashes.json profiles/${RD}_kernel_hashes.json
echo "kernel hashes of the library in this tree:"; python3 tools/kernel_hash.py | grep -E "k_move_tt|k_tail|k_res|k_move_t\""; grep -h kernel_hash profiles/pmc_*.json
