# per-rank step (bench.py --samples S) against the work-item count of the matrix-pipe reverse pass (probe build knob MXF_BWD_MFMA_GRID)
cd $GRAFT_REPO_ROOT
export MXF_GP_LIB=$PWD/mxfusion_amd/libmxf_gp_probe.so
for S in ${1:-4}; do for g in 8192 4096 2048 1024 512; do
  echo -n "S=$S grid=$g  "
  MXF_BWD_MFMA_GRID=$g python bench.py --steps 30 --warmup 5 --samples $S --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"
done; done
