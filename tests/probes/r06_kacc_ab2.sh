#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s4
bash tests/probes/r06_gp_ab.sh "MXF_POTRF_KACC=0,MXF_POTRF_CHAIN_PRIO=0 MXF_POTRF_KACC=0,MXF_POTRF_CHAIN_PRIO=1 MXF_POTRF_KACC=1,MXF_POTRF_CHAIN_PRIO=1" > gpurun_out/s4/kacc_ab2.txt 2>&1
cat gpurun_out/s4/kacc_ab2.txt
