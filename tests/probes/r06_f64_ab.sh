# float64 step time for probe-build settings: r06_f64_ab.sh "ENV=VAL ..."
cd $GRAFT_REPO_ROOT
export MXF_GP_LIB=$PWD/mxfusion_amd/libmxf_gp_probe.so
for cfg in $1; do
  echo -n "f64 $cfg  "
  env $(echo $cfg | tr ',' ' ') python bench.py --dtype float64 --steps 4 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d['last_loss'])"
done
