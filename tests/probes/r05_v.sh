#!/bin/bash
# r05: the whitened tier's V product with paired strips (ascending / descending k) against the r04 rotation: time, fabric fetch, step
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_v
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_whitened.py -x -q > $O/pytest_whitened.log 2>&1; tail -3 $O/pytest_whitened.log
export MXF_GP_LIB=$R/mxfusion_amd/libmxf_gp_probe.so
for rep in 1 2; do
  for pair in 0 1; do
    echo "== MXF_SPLIT_PAIR=$pair rep $rep"; MXF_SPLIT_PAIR=$pair python tests/probes/vgemm_time.py
  done
done 2>&1 | tee $O/vgemm_time.log
echo "== pair, no rendezvous"; MXF_SPLIT_PAIR=1 MXF_SPLIT_SYNC=0 python tests/probes/vgemm_time.py 2>&1 | tee -a $O/vgemm_time.log
cd /tmp && export TMPDIR=/tmp
for pair in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    MXF_SPLIT_PAIR=$pair rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_v${pair}_$c -o g -- python $R/tests/probes/split_pmc.py v > $O/pmc_v${pair}_$c.log 2>&1
  done
  MXF_SPLIT_PAIR=$pair rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_v${pair}_sq -o g -- python $R/tests/probes/split_pmc.py v > $O/pmc_v${pair}_sq.log 2>&1
  python $R/profiles/pmc_summary.py gemm_f16x2 $O/gemm_v${pair}_pmc.json $O/pmc_v${pair}_FETCH_SIZE $O/pmc_v${pair}_WRITE_SIZE $O/pmc_v${pair}_sq > $O/gemm_v${pair}_pmc.txt 2>&1
  cat $O/gemm_v${pair}_pmc.txt
done
unset MXF_GP_LIB
cd $R
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --trained-like > $O/bench_trained_like.json 2> $O/bench_trained_like.err; cat $O/bench_trained_like.json
