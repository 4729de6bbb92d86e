# same-box A/B of the exact-GP float64 step (configs[1]) under one environment knob:  gp_ab.sh VAR a b
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in $2 $3; do
  echo -n "$1=$v  "
  env $1=$v timeout 300 python bench.py --workload gp --dtype float64 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"
done; done
