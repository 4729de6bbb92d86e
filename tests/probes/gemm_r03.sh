#!/bin/bash
# r03: A/B of the wide split GEMM variants (rows per tile XT, rendezvous mode SYNC) on the T and Psi2 shapes -- standalone time
# (t_time.py) and fabric fetch per launch (rocprofv3 --pmc FETCH_SIZE; gfx950: x 2 x 1024 for bytes).  Probe build of the library.
# usage: gemm_r03.sh time|pmc "<xt>:<sync>[:<pp>] ..."   (xt 4 | 8 | 16 = 256 rows by eight waves; pp = ping-pong phases)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/gemm_r03
mkdir -p $O
export MXF_GP_LIB=$R/mxfusion_amd/libmxf_gp_probe.so
cd /tmp && export TMPDIR=/tmp
mode=$1; shift
for cfg in $1; do
  IFS=: read xt sy pp <<< "$cfg"; pp=${pp:-0}
  export MXF_SPLIT_XT=$xt MXF_SPLIT_SYNC=$sy MXF_SPLIT_PP=$pp
  if [ "$mode" = time ]; then
    echo "== XT=$xt SYNC=$sy PP=$pp"; python $R/tests/probes/t_time.py 2>&1 | tail -2
  else
    for w in t psi2; do
      d=$O/pmc_${w}_xt${xt}_s${sy}_pp${pp}
      rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $d -o g -- python $R/tests/probes/split_pmc.py $w > $d.log 2>&1
      echo "== pmc $w XT=$xt SYNC=$sy PP=$pp"; python $R/profiles/pmc_summary.py gemm_f16x2 $d.json $d | grep -v "^_kernel"
      python - $d <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in csv.DictReader(open(f)) if 'gemm_f16x2' in r['Kernel_Name']]
    print('   kernel ms (profiled):', ['%.2f' % x for x in d])
PY
    done
  fi
done
