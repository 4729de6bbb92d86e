cd $GRAFT_REPO_ROOT
export MXF_GP_LIB=$PWD/mxfusion_amd/libmxf_gp_probe.so
for v in 0 1 0 1; do MXF_BWD_DMAT=$v python bench.py --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DMAT=$v', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d.get('step_breakdown_ms',{}).items()})"; done
