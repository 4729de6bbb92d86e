#!/bin/bash
# r06: the planes pass as w persistent waves per CU (MXF_PLANES_PERSIST), so that the Kuu chain's workgroups find room next to it
cd $GRAFT_REPO_ROOT
O=gpurun_out/s4; mkdir -p $O
export MXF_GP_LIB=$PWD/mxfusion_amd/libmxf_gp_probe.so
MXF_PLANES_PERSIST=28 python -m pytest tests/test_gpu_whitened.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_gram.py -x -q 2>&1 | grep -E "passed|failed" > $O/persist.txt
for rep in 1 2; do for w in 0 28 24 16; do
  for args in "--samples 32 --trained-like" "--samples 32" "--samples 4 --trained-like"; do
    echo -n "PERSIST=$w rep=$rep $args: "
    MXF_PLANES_PERSIST=$w python bench.py $args --steps 20 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d.get('ms_per_step_trained_like'))"
  done
done; done >> $O/persist.txt 2>&1
cat $O/persist.txt
