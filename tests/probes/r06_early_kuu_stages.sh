#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s4; mkdir -p $O
for k in 0 1 0 1; do
  echo "=== EARLY_KUU=$k, 4 samples"; MXF_SVGP_EARLY_KUU=$k python tests/probes/svgp_stages.py 4
done > $O/early_stages.txt 2>&1
for k in 0 1; do
  echo "=== EARLY_KUU=$k, 4 samples minibatch 8192"; MXF_SVGP_EARLY_KUU=$k python tests/probes/svgp_stages.py 4 --minibatch 8192
done >> $O/early_stages.txt 2>&1
cat $O/early_stages.txt
