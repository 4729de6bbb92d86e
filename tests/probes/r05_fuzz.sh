#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_fuzz; mkdir -p $O; cd $R
for seed in 31 32; do timeout 1200 python tests/probes/fuzz_svgp.py 80 $seed 2>&1 | grep -v "amdgpu" > $O/fuzz_svgp_$seed.log; echo "seed $seed: $(grep -c BAD $O/fuzz_svgp_$seed.log) BAD of $(grep -c 'f32-f64' $O/fuzz_svgp_$seed.log)"; grep BAD $O/fuzz_svgp_$seed.log | head -10; done
timeout 900 python tests/probes/fuzz_svgp_comb.py 40 31 2>&1 | grep -v amdgpu > $O/fuzz_comb_31.log; echo "comb: $(grep -c BAD $O/fuzz_comb_31.log) BAD"; grep BAD $O/fuzz_comb_31.log | head -5
