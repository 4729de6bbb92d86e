#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_deepgp; mkdir -p $O; cd $R
for i in 1 2 3; do python -m pytest tests/test_gpu_more_api.py -q -x -k two_layer 2>&1 | tail -1; done | tee $O/pytest.log
for rep in 1 2; do for flag in "--serial-modules" ""; do for s in 32 4; do
  echo "deepgp S=$s ${flag:-concurrent}: $(python bench.py --workload deepgp --samples $s $flag --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],3), d['last_loss'], d['float32_tiers'])")"
done; done; done 2>&1 | tee $O/deepgp.log
