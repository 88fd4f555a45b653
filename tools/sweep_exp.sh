# bench lines of the experiment libraries under traffic-env_amd/lib/exp (make -C traffic-env_amd/csrc exp EXP=...), next to the default one
cd $GRAFT_REPO_ROOT
for l in default $(ls traffic-env_amd/lib/exp | sed 's/libtfx_//; s/.so//') default; do
  if [ $l = default ]; then unset TFX_LIB; else export TFX_LIB=$GRAFT_REPO_ROOT/traffic-env_amd/lib/exp/libtfx_$l.so; fi
  python3 bench.py --steps ${STEPS:-200} --warmup 20 --repeats 3 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib $l', round(d['ms_per_step'],4), '%.4g'%d['value'], 'launch', round(d['roofline']['launch_ms'],4), 'agent', d.get('agent_decision_ms') and round(d['agent_decision_ms'],3))"
done
