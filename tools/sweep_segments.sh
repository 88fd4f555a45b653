#!/bin/bash
# k_move_tts (every tile's walk of a two-tick pass split over S wavefronts): the sweep behind tt_segments / pairs_usable.
#   bash tools/sweep_segments.sh > gpurun_out/segments.txt      (run through gpurun)
cd $GRAFT_REPO_ROOT
c4() { python3 tools/c4_loop.py $1 2>&1 | grep closed | sed 's/, [0-9]* cars.*//'; }
b2() { python3 bench.py --config $1 --envs $2 --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%s x %s: %.1f us per tick (%s), fused 10-tick decision %.1f us' % ('$1', '$2', d['ms_per_step']*1e3, d['roofline']['kernel'], d['agent_decision_ms']*1e3))"; }
echo "# cfg4 closed loop (64x64, 128-car rings, on-device Poisson + greedy, empty start), env-ticks/s; S = segments per tile"
for E in 1 2 4 8 16; do
  echo "default:            $(c4 $E)"
  echo "tick by tick:       $(TFX_PAIRS=0 c4 $E)"
  echo "pairs, one wavefront per tile: $(TFX_PAIRS=2 TFX_TT_SEG=0 c4 $E)"
  for S in 2 4 8; do echo "pairs, S = $S:       $(TFX_PAIRS=2 TFX_TT_SEG=2 TFX_TT_SEGS=$S c4 $E)"; done
done
echo "# cfg2 (16x16, 64-car rings), the benchmark's workload: the handle's choice | tick by tick | pairs without segments"
for E in 1 8 32 64 128 256; do
  echo "default:      $(b2 cfg2 $E)"
  echo "tick by tick: $(TFX_PAIRS=0 b2 cfg2 $E)"
  echo "S = 0:        $(TFX_PAIRS=2 TFX_TT_SEG=0 b2 cfg2 $E)"
done
echo "# cfg4 prefilled (every road at 96 of 128 cars)"
for E in 1 4 16; do
  echo "default:      $(b2 cfg4 $E)"
  echo "tick by tick: $(TFX_PAIRS=0 b2 cfg4 $E)"
  echo "S = 0:        $(TFX_PAIRS=2 TFX_TT_SEG=0 b2 cfg4 $E)"
done
