# T-product traffic and step time against the rendezvous period of gemm_bt.hip (probe build): r06_bt_sync.sh "0 1 4 8 16"
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp && export TMPDIR=/tmp; export MXF_GP_LIB=$R/mxfusion_amd/libmxf_gp_probe.so
O=$R/gpurun_out/bt_sync; mkdir -p $O
for v in $1; do
  export MXF_BT_SYNC=$v
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f_$v -o g -- python $R/tests/probes/split_pmc.py tbt > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w_$v -o g -- python $R/tests/probes/split_pmc.py tbt > /dev/null 2>&1
  python $R/profiles/pmc_summary.py gemm_f16x2 $O/pmc_$v.json $O/f_$v $O/w_$v > $O/pmc_$v.txt 2>&1
  echo -n "MXF_BT_SYNC=$v  traffic $(grep hbm_traffic $O/pmc_$v.txt)  step "
  (cd $R; python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))")
  rm -rf $O/f_$v $O/w_$v
done
