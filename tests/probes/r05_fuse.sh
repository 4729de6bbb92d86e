#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_fuse
mkdir -p $O
cd $R
timeout 600 python tests/probes/fuse_check.py 2>&1 | grep -v amdgpu | tee $O/fuse_check.log
export MXF_GP_LIB=$R/mxfusion_amd/libmxf_gp_probe.so
for rep in 1 2; do for f in 0 1; do
  for extra in "" "--trained-like"; do
    echo "fuse=$f $extra: $(MXF_SVGP_FUSE=$f timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras $extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],3), d.get('last_loss'))")"
  done
  echo "fuse=$f S=4: $(MXF_SVGP_FUSE=$f timeout 300 python bench.py --steps 40 --warmup 5 --samples 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],3))")"
done; done 2>&1 | tee $O/fuse_time.log
MXF_SVGP_FUSE=1 timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print({k: d.get(k) for k in ('ms_per_step','step_breakdown_ms','ms_per_step_trained_like','elbo_f32_vs_f64_rel','elbo_f32_vs_f64_rel_trained_like')})" | tee $O/fuse_breakdown.log
