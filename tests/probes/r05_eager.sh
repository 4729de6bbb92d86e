#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_eager; mkdir -p $O; cd $R
export MXF_GP_LIB=$R/mxfusion_amd/libmxf_gp_probe.so
for rep in 1 2; do for cfg in "0 0" "4 0" "4 64" "4 128" "2 64" "3 64" "8 0"; do set -- $cfg
  echo "eager=$1 res=$2: $(MXF_POTRF_EAGER_INV=$1 MXF_POTRF_EAGER_RES=$2 python bench.py --workload gp --dtype float64 --N 8192 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],3))")"
done; done 2>&1 | tee $O/eager3.log
