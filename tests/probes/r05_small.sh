#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_small2; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_minibatch_graph.py -x -q 2>&1 | tail -5 | tee $O/pytest.log
timeout 900 python tests/probes/small_run.py 10 2>&1 | grep -v "amdgpu\|Warning\|warn" | tee $O/small_run.log
