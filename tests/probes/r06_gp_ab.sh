# exact-GP MAP step for probe-build settings, alternating: r06_gp_ab.sh "ENV=V,ENV=V ..."
cd $GRAFT_REPO_ROOT
export MXF_GP_LIB=$PWD/mxfusion_amd/libmxf_gp_probe.so
for rep in 1 2; do for cfg in $1; do
  echo -n "$cfg rep=$rep  "
  env $(echo $cfg | tr ',' ' ') python bench.py --workload gp --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MAP step', round(d['ms_per_step'],3), d.get('last_loss'))"
done; done
