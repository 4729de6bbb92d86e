# rows below an outer panel in workgroups of 64 / 32 / 16 rows (MXF_POTRF_ROWS_RH, probe build): potrf(n) and the exact-GP MAP step, alternating
cd $GRAFT_REPO_ROOT
export MXF_GP_LIB=$PWD/mxfusion_amd/libmxf_gp_probe.so
mkdir -p gpurun_out/rows_rh
for rep in 1 2; do for rh in 64 32 16; do
  echo "== RH=$rh rep=$rep"
  MXF_POTRF_ROWS_RH=$rh python tests/probes/potrf_time.py 2>/dev/null | tr '\n' '|'; echo
  MXF_POTRF_ROWS_RH=$rh python bench.py --workload gp --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MAP step', round(d['ms_per_step'],3), d.get('last_loss'))"
done; done 2>&1 | tee gpurun_out/rows_rh/ab.log
MXF_POTRF_ROWS_RH=32 python -m pytest tests/test_gpu_linalg.py -q -x 2>&1 | tail -3
