# mid batch sizes of cfg2: the handle's own choices against forced k_tail / split (us per tick)
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --config cfg2 --envs $1 --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline --no-agent-steps 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2 x $1 $2', round(d['ms_per_step']*1e3,1), 'us/tick', d['roofline']['kernel'], 'split' if d.get('ticks_split_over_two_streams') else '', 'tail' if d.get('ticks_finished_by_k_tail', d.get('tail_ticks')) else '')"; }
for E in ${ENVS:-96 128 192 256 384 512 768}; do
  unset TFX_TAIL TFX_SPLIT; run $E default
  TFX_TAIL=2 run $E tail=2
  TFX_TAIL=2 TFX_SPLIT=2 run $E tail=2,split=2
  TFX_SPLIT=0 run $E split=0
done
