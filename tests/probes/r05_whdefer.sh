#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_whdefer; mkdir -p $O; cd $R
export MXF_GP_LIB=$R/mxfusion_amd/libmxf_gp_probe.so
t() { timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.readlines()[-1])['ms_per_step'],3))"; }
for rep in 1 2; do for d in 0 1 2; do
  echo "rep $rep defer=$d  S=32 trained-like: $(MXF_SVGP_WH_DEFER=$d t --trained-like --steps 10 --warmup 3)  S=4 trained-like: $(MXF_SVGP_WH_DEFER=$d t --samples 4 --trained-like --steps 40 --warmup 5)"
done; done 2>&1 | tee $O/whdefer.log
