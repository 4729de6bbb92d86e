#!/bin/bash
# r03: the split-plane Gram passes -- staged 256-thread kernel (MXF_PLANES_LEAN=0) against the one-wave scalar-load form, and the
# k blocks per workgroup of the latter -- step time at 32 samples and kernel times from a rocprofv3 kernel trace.
# usage: bash tests/probes/planes_lean.sh "<lean>:<kb> ..."
export MXF_GP_LIB=$PWD/mxfusion_amd/libmxf_gp_probe.so
mkdir -p gpurun_out/planes
for cfg in ${1:-"0:8 1:8"}; do
  IFS=: read lean kb <<< "$cfg"
  export MXF_PLANES_LEAN=$lean MXF_PLANES_KB=$kb
  ms=$(python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])")
  echo "lean=$lean kb=$kb step $ms"
  ROOT=$PWD
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/planes/l${lean}_k${kb} -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $ROOT/gpurun_out/planes/l${lean}_k${kb}.log 2>&1)
  python - <<P
import csv,glob
f=glob.glob('gpurun_out/planes/l${lean}_k${kb}/**/*kernel_stats.csv',recursive=True)
if not f: print('   no stats file', glob.glob('gpurun_out/planes/l${lean}_k${kb}/**', recursive=True)[:5])
for r in csv.DictReader(open(f[0])) if f else []:
    if 'planes' in r['Name']: print('   ', r['Name'][:70], r['Calls'], r['AverageNs'])
P
done
