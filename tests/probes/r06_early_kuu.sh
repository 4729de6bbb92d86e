#!/bin/bash
# r06 session 4: the R-independent part of dKuu on the side stream (MXF_SVGP_EARLY_KUU): GPU suite, then same-box A/B through the probe build
cd $GRAFT_REPO_ROOT
O=gpurun_out/s4; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > $O/early_tests.txt
export MXF_GP_LIB=$PWD/mxfusion_amd/libmxf_gp_probe.so
for rep in 1 2; do for k in 0 1; do
  for args in "--samples 32" "--samples 4" "--samples 4 --trained-like" "--minibatch 8192 --samples 4" "--minibatch 8192"; do
    echo -n "EARLY_KUU=$k rep=$rep $args: "
    MXF_SVGP_EARLY_KUU=$k python bench.py $args --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"
  done
done; done > $O/early_ab.txt 2>&1
cat $O/early_tests.txt $O/early_ab.txt
