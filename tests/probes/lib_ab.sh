# same-box A/B of two builds of the library on the SVGP step:  lib_ab.sh libA.so libB.so   (paths relative to mxfusion_amd/)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for l in $1 $2; do
  echo -n "$l  "
  for args in "--samples 32" "--samples 4" "--minibatch 8192 --samples 4"; do
    MXF_GP_LIB=$PWD/mxfusion_amd/$l timeout 300 python bench.py --steps 20 --warmup 5 $args --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), end='  ')"
  done; echo
done; done
