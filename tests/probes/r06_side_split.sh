#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s4; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | grep -E "FAILED|^E |passed|failed" | head -20 > $O/side_split.txt
export MXF_GP_LIB=$PWD/mxfusion_amd/libmxf_gp_probe.so
for rep in 1 2 3; do for k in 0 1; do
  for args in "--samples 4" "--samples 4 --trained-like" "--minibatch 8192 --samples 4"; do
    echo -n "SIDE_SPLIT=$k rep=$rep $args: "
    MXF_SVGP_SIDE_SPLIT=$k python bench.py $args --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"
  done
done; done >> $O/side_split.txt 2>&1
for k in 0 1; do echo "=== SIDE_SPLIT=$k"; MXF_SVGP_SIDE_SPLIT=$k python tests/probes/svgp_stages.py 4 | grep -E "Su reverse|reverse pass|core reverse|end"; done >> $O/side_split.txt 2>&1
cat $O/side_split.txt
