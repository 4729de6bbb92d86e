#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_b3
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -25 $O/pytest_gpu.log
python tests/probes/vgemm_time.py 2>&1 | grep -v amdgpu | tee $O/vgemm_time.log
for rep in 1 2; do for p in 0 1; do echo "prio=$p"; MXF_STREAM_PRIO=$p MXF_GP_LIB=$R/mxfusion_amd/libmxf_gp_probe.so python bench.py --steps 40 --warmup 5 --samples 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.readlines()[-1])['ms_per_step'],3))"; done; done 2>&1 | tee $O/prio.log
