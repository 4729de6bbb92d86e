# step time for a list of "ENV=VAL,ENV=VAL" settings of the probe build (same box, alternating): r06_ab.sh SAMPLES "cfg1 cfg2 ..."
cd $GRAFT_REPO_ROOT
export MXF_GP_LIB=$PWD/mxfusion_amd/libmxf_gp_probe.so
S=${1:-32}; shift
for cfg in $1; do
  echo -n "S=$S $cfg  "
  env $(echo $cfg | tr ',' ' ') python bench.py --steps 20 --warmup 4 --samples $S --no-cpu-baseline --no-extras ${EXTRA} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d['last_loss'])"
done
