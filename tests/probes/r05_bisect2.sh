#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_bisect; mkdir -p $O; cd $R
run() { name=$1; shift; python -m pytest "$@" tests/test_gpu_minibatch_graph.py -q -x -v > $O/$name.log 2>&1; echo "$name: rc=$? segv=$(grep -c 'Segmentation' $O/$name.log) $(grep -E 'test_gpu_minibatch_graph.*(PASSED|FAILED)|passed|failed' $O/$name.log | tail -3 | tr '\n' ' ' | cut -c1-300)"; }
run A tests/test_gpu_fullsize.py tests/test_gpu_fullsize_oracle.py
run B tests/test_gpu_config4.py tests/test_gpu_edges.py tests/test_gpu_customop.py tests/test_gpu_comm.py
run C tests/test_gpu_gemm.py tests/test_gpu_gp_dist.py tests/test_gpu_inference_alg.py
run D tests/test_gpu_api.py tests/test_gpu_comm.py tests/test_gpu_composites.py tests/test_gpu_config4.py tests/test_gpu_customop.py tests/test_gpu_edges.py tests/test_gpu_f32_guard.py
