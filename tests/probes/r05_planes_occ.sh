#!/bin/bash
# r05: occupancy-limited planes pass (MXF_PLANES_WAVES = 4 / 5 / 6 waves per SIMD, compile-time variants) against the shipped form (8)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_occ
mkdir -p $O
cd $R
t() { MXF_GP_LIB=$R/mxfusion_amd/$1 timeout 300 python bench.py --no-cpu-baseline --no-extras "${@:2}" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.readlines()[-1])['ms_per_step'],3))"; }
for rep in 1 2; do for lib in libmxf_gp.so libmxf_gp_w4.so libmxf_gp_w5.so libmxf_gp_w6.so; do
  echo "rep $rep $lib  S=4: $(t $lib --samples 4 --steps 40 --warmup 5)  S=4 trained-like: $(t $lib --samples 4 --trained-like --steps 40 --warmup 5)  S=32: $(t $lib --steps 10 --warmup 3)  S=32 trained-like: $(t $lib --trained-like --steps 10 --warmup 3)  mb8192 S=4: $(t $lib --minibatch 8192 --samples 4 --steps 40 --warmup 5)"
done; done 2>&1 | tee $O/occ.log
