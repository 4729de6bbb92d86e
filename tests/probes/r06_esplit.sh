# eager inverse in split pieces (MXF_POTRF_EAGER_SPLIT) x row-block granularity (MXF_POTRF_EAGER_INV / _TAIL), probe build: exact-GP MAP step, alternating
cd $GRAFT_REPO_ROOT
export MXF_GP_LIB=$PWD/mxfusion_amd/libmxf_gp_probe.so
mkdir -p gpurun_out/esplit
for rep in 1 2; do for cfg in "MXF_POTRF_EAGER_SPLIT=0" "MXF_POTRF_EAGER_SPLIT=1" "MXF_POTRF_EAGER_INV=2" "MXF_POTRF_EAGER_INV=3" "MXF_POTRF_EAGER_INV=2,MXF_POTRF_EAGER_TAIL=1" "MXF_POTRF_EAGER_INV=4,MXF_POTRF_EAGER_TAIL=2" "MXF_POTRF_EAGER_INV=2,MXF_POTRF_ROWS_RH=64"; do
  echo -n "$cfg rep=$rep  "
  env $(echo $cfg | tr ',' ' ') python bench.py --workload gp --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MAP step', round(d['ms_per_step'],3), d.get('last_loss'))"
done; done 2>&1 | tee gpurun_out/esplit/ab3.log
