#!/bin/bash
# r06 session 4: -P/2 L^-T L^-1 accumulated next to the factorisation (MXF_POTRF_KACC) and CU-masked bulk streams (MXF_POTRF_CUMASK = CUs per XCD left to the chain)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s4
python -m pytest tests/test_gpu_linalg.py tests/test_gpu_fullsize.py tests/test_gpu_api.py -x -q 2>&1 | tail -3 > gpurun_out/s4/kacc_tests.txt
bash tests/probes/r06_gp_ab.sh "MXF_POTRF_KACC=0,MXF_POTRF_CUMASK=0 MXF_POTRF_KACC=1,MXF_POTRF_CUMASK=0 MXF_POTRF_KACC=0,MXF_POTRF_CUMASK=4 MXF_POTRF_KACC=1,MXF_POTRF_CUMASK=4 MXF_POTRF_KACC=1,MXF_POTRF_CUMASK=2 MXF_POTRF_KACC=1,MXF_POTRF_CUMASK=8" > gpurun_out/s4/kacc_ab.txt 2>&1
cat gpurun_out/s4/kacc_tests.txt gpurun_out/s4/kacc_ab.txt
