# Sweep of the two-tick pass's grid (workgroups per CU) for the split call: ms/tick, veh-upd/s, pass launch ms, agent decision ms
cd $GRAFT_REPO_ROOT
for b in ${BLOCKS:-0 15 16 17 0 16 17}; do
  if [ $b = 0 ]; then unset TFX_MOVE_BLOCKS_PER_CU; else export TFX_MOVE_BLOCKS_PER_CU=$b; fi
  python3 bench.py --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('blocks/CU $b', round(d['ms_per_step'],4), '%.4g'%d['value'], round(d['roofline']['launch_ms'],4), 'agent', round(d['agent_decision_ms'],3))"
done
