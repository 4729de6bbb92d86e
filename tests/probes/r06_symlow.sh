#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s4; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > $O/symlow.txt
export MXF_GP_LIB=$PWD/mxfusion_amd/libmxf_gp_probe.so
for rep in 1 2 3; do for k in 0 1; do
  for args in "--samples 4" "--minibatch 8192 --samples 4"; do
    echo -n "SYM_LOWER=$k rep=$rep $args: "
    MXF_SVGP_SYM_LOWER=$k python bench.py $args --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"
  done
done; done >> $O/symlow.txt 2>&1
for k in 0 1; do echo "=== SYM_LOWER=$k"; MXF_SVGP_SYM_LOWER=$k python tests/probes/svgp_stages.py 4 | grep -E "trtri|Ki, w|H0 planes|end"; done >> $O/symlow.txt 2>&1
cat $O/symlow.txt
