#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_fuse2
mkdir -p $O
cd $R
timeout 600 python tests/probes/fuse_check.py 2>&1 | grep -v amdgpu | grep -A3 "M=512\|M=1024" | tee $O/fuse_check.log
export MXF_GP_LIB=$R/mxfusion_amd/libmxf_gp_probe.so
for f in 0 1; do
  echo "fuse=$f: $(MXF_SVGP_FUSE=$f timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],3), d.get('last_loss'))")"
done 2>&1 | tee $O/fuse_time.log
MXF_SVGP_FUSE=1 timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print({k: d.get(k) for k in ('ms_per_step','step_breakdown_ms')})" | tee $O/fuse_breakdown.log
unset MXF_GP_LIB
bash tests/probes/step_listing.sh s4 --samples 4 > /dev/null 2>&1
cp gpurun_out/listing_s4.txt $O/ 2>/dev/null
