#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s4; mkdir -p $O
export MXF_GP_LIB=$PWD/mxfusion_amd/libmxf_gp_probe.so
python -m pytest tests/test_gpu_sweep.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_api.py -x -q 2>&1 | grep -E "passed|failed" > $O/ra.txt
for rep in 1 2; do for ra in 206 214 222 230; do
  for args in "--samples 4" "--minibatch 8192 --samples 4"; do
    echo -n "RA=$ra rep=$rep $args: "
    MXF_SVGP_PSI2_RA=$ra python bench.py $args --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"
  done
done; done >> $O/ra.txt 2>&1
cat $O/ra.txt
