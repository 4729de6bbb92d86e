#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s4; mkdir -p $O
python -m pytest tests/test_gpu_gemm.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_composites.py tests/test_gpu_training_f32.py tests/test_gpu_whitened.py -x -q 2>&1 | grep -E "passed|failed|error" > $O/het_split2.txt
export MXF_GP_LIB=$PWD/mxfusion_amd/libmxf_gp_probe.so
for rep in 1 2; do for k in 0 1; do
  for args in "--samples 32 --steps 10" "--samples 4 --steps 20"; do
    echo -n "HET_SPLIT=$k rep=$rep deepgp $args: "
    MXF_SVGP_HET_SPLIT=$k python bench.py --workload deepgp $args --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d.get('last_loss'))"
  done
done; done >> $O/het_split2.txt 2>&1
cat $O/het_split2.txt
