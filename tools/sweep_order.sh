# the split call: submission order of the halves / stagger, at the driver's settings and at 200 ticks
cd $GRAFT_REPO_ROOT
for st in 1 0 1 0; do
  export TFX_STAGGER=$st
  for steps in "20 5" "200 20"; do
    set -- $steps
    python3 bench.py --steps $1 --warmup $2 --repeats 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('stagger $st steps $1', round(d['ms_per_step'],4), '%.4g'%d['value'], [round(x,4) for x in d['ms_per_step_spread']], 'agent', round(d['agent_decision_ms'],3))"
  done
done
