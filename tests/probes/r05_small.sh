#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_small; mkdir -p $O; cd $R
(timeout 600 python tests/probes/small_step.py 1000 20 10 1; timeout 600 python tests/probes/small_step.py 65536 1024 1024 8; timeout 600 python tests/probes/small_step.py 65536 1024 8192 8) 2>&1 | grep -v amdgpu | tee $O/small_step.log
