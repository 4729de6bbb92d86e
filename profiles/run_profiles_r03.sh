#!/bin/bash
# Round-3 profiles (run on the GPU box through gpurun, from the repo root); summaries are copied into profiles/ by hand afterwards.
# Counters are collected in their own passes with --kernel-trace only (never with sys / hip / hsa traces).
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. the default bench command with its roofline micro-measurements: per-kernel averages (the gram_lean_kernel<float, 8, 0> row is the
#    kernel bench.py's `roofline` times with HIP events), and one step as a timeline
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_full -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_trace_full.log 2>&1
# (the timeline tool shows the LAST step of a trace: the step alone, without the roofline extras that follow it in the full command)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_step -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_trace_step.log 2>&1
python $R/profiles/timeline.py $(find $O/trace_step -name "*kernel_trace.csv") 0.15 > $O/step_timeline.txt 2>&1
# 2. the per-rank share of an 8-GPU run (4 samples) as a timeline: the core chain is the critical path there
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_s4 -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --samples 4 > $O/bench_trace_s4.log 2>&1
python $R/profiles/timeline.py $(find $O/trace_s4 -name "*kernel_trace.csv") 0.03 > $O/step_timeline_s4.txt 2>&1
# 3. PMC passes of the Gram kernel (HBM traffic per launch)
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_gram_write -o gram -- python $R/tests/probes/gram_f32.py > $O/pmc_gram_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_gram_fetch -o gram -- python $R/tests/probes/gram_f32.py > $O/pmc_gram_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_gram_sq -o gram -- python $R/tests/probes/gram_f32.py > $O/pmc_gram_sq.log 2>&1
python $R/profiles/pmc_summary.py gram_lean_kernel $O/gram_pmc.json $O/pmc_gram_write $O/pmc_gram_fetch $O/pmc_gram_sq > $O/gram_pmc.txt 2>&1
# 4. PMC passes of the two split GEMMs
for w in t psi2; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_${w}_fetch -o g -- python $R/tests/probes/split_pmc.py $w > $O/pmc_${w}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_${w}_write -o g -- python $R/tests/probes/split_pmc.py $w > $O/pmc_${w}_write.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${w}_sq -o g -- python $R/tests/probes/split_pmc.py $w > $O/pmc_${w}_sq.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $O/pmc_${w}_lds -o g -- python $R/tests/probes/split_pmc.py $w > $O/pmc_${w}_lds.log 2>&1
  python $R/profiles/pmc_summary.py gemm_ $O/gemm_${w}_pmc.json $O/pmc_${w}_fetch $O/pmc_${w}_write $O/pmc_${w}_sq $O/pmc_${w}_lds > $O/gemm_${w}_pmc.txt 2>&1
done
# 5. PMC passes of the matrix-pipe reverse pass (inside the bench step)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_bwd_sq -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_bwd_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES --output-format csv -d $O/pmc_bwd_lds -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_bwd_lds.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_bwd_fetch -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_bwd_fetch.log 2>&1
python $R/profiles/pmc_summary.py svgp_bwd_mfma $O/bwd_pmc.json $O/pmc_bwd_sq $O/pmc_bwd_lds $O/pmc_bwd_fetch > $O/bwd_pmc.txt 2>&1
ls -R $O | head -60
