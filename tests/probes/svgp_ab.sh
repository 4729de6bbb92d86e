# same-box A/B of the SVGP step under one environment knob:  svgp_ab.sh VAR a b   (32 samples, 4 samples, minibatch 8192 at 4 samples)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in $2 $3 $4; do
  echo -n "$1=$v  "
  for args in "--samples 32" "--samples 4" "--minibatch 8192 --samples 4"; do
    env $1=$v timeout 300 python bench.py --steps 20 --warmup 5 $args --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), end='  ')"
  done; echo
done; done
