#!/bin/bash
# Round-6 PMC passes of the T product: the r05 kernel (row-operand planes, B fragments straight from global memory) against gemm_bt.hip
# (K-major planes through LDS).  Counters in their own passes with --kernel-trace only.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in ${1:-t tbt tbt0}; do
  k=gemm_f16x2
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_${w}_fetch -o g -- python $R/tests/probes/split_pmc.py $w > $O/pmc_${w}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_${w}_write -o g -- python $R/tests/probes/split_pmc.py $w > $O/pmc_${w}_write.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${w}_sq -o g -- python $R/tests/probes/split_pmc.py $w > $O/pmc_${w}_sq.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_${w}_lds -o g -- python $R/tests/probes/split_pmc.py $w > $O/pmc_${w}_lds.log 2>&1
  python $R/profiles/pmc_summary.py $k $O/gemm_${w}_pmc.json $O/pmc_${w}_fetch $O/pmc_${w}_write $O/pmc_${w}_sq $O/pmc_${w}_lds > $O/gemm_${w}_pmc.txt 2>&1
done
rm -rf $O/pmc_*_fetch $O/pmc_*_write $O/pmc_*_sq $O/pmc_*_lds
ls $O
