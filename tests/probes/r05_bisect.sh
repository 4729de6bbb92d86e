#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_bisect; mkdir -p $O; cd $R
for pre in test_gpu_api.py test_gpu_f32_guard.py test_gpu_mean_function.py test_gpu_linalg.py test_gpu_gram.py test_gpu_composites.py; do
  python -m pytest tests/$pre tests/test_gpu_minibatch_graph.py -q -x > $O/$pre.log 2>&1; echo "$pre + graph: rc=$? $(grep -c 'Segmentation' $O/$pre.log) segv; $(tail -1 $O/$pre.log | cut -c1-80)"
done
