#!/bin/bash
# Round-5 profiles (run on the GPU box through gpurun, from the repo root); summaries are copied into profiles/ afterwards.
# Counters are collected in their own passes with --kernel-trace only (never with sys / hip / hsa traces).
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_r05
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. the default bench command with its roofline micro-measurements: per-kernel averages (the gram_lean_kernel<float, 8, 0> row is the kernel
#    bench.py's `roofline` times with HIP events), and one step as a timeline
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_full -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_trace_full.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_step -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_trace_step.log 2>&1
python $R/profiles/timeline.py $(find $O/trace_step -name "*kernel_trace.csv") 0.15 > $O/step_timeline.txt 2>&1
# 2. the step in the WHITENED float32 form at trained-like parameters (what the guard selects above cond 1e3): kernel averages + timeline
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_whitened -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --trained-like > $O/bench_trace_whitened.log 2>&1
python $R/profiles/timeline.py $(find $O/trace_whitened -name "*kernel_trace.csv") 0.15 > $O/step_timeline_whitened.txt 2>&1
# 3. the per-rank share of an 8-GPU run (4 samples) as a timeline
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_s4 -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --samples 4 > $O/bench_trace_s4.log 2>&1
python $R/profiles/timeline.py $(find $O/trace_s4 -name "*kernel_trace.csv") 0.03 > $O/step_timeline_s4.txt 2>&1
# 4. PMC passes of the Gram kernel (HBM traffic per launch)
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_gram_write -o gram -- python $R/tests/probes/gram_f32.py > $O/pmc_gram_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_gram_fetch -o gram -- python $R/tests/probes/gram_f32.py > $O/pmc_gram_fetch.log 2>&1
python $R/profiles/pmc_summary.py gram_lean_kernel $O/gram_pmc.json $O/pmc_gram_write $O/pmc_gram_fetch > $O/gram_pmc.txt 2>&1
# 5. PMC passes of the split GEMMs: t / psi2 as before; v = the whitened tier's triangular planes-output product as the step runs it (r05:
#    paired column strips, ascending / descending k: the strip's B planes are fetched once instead of once per row tile)
for w in t psi2 v; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_${w}_fetch -o g -- python $R/tests/probes/split_pmc.py $w > $O/pmc_${w}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_${w}_write -o g -- python $R/tests/probes/split_pmc.py $w > $O/pmc_${w}_write.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${w}_sq -o g -- python $R/tests/probes/split_pmc.py $w > $O/pmc_${w}_sq.log 2>&1
  python $R/profiles/pmc_summary.py gemm_f16x2 $O/gemm_${w}_pmc.json $O/pmc_${w}_fetch $O/pmc_${w}_write $O/pmc_${w}_sq > $O/gemm_${w}_pmc.txt 2>&1
done
for d in trace_full trace_step trace_whitened trace_s4; do cp $(find $O/$d -name "*kernel_stats.csv" | head -1) $O/stats_${d}.csv; done
rm -rf $O/trace_* $O/pmc_*_fetch $O/pmc_*_write $O/pmc_*_sq $O/pmc_gram_write $O/pmc_gram_fetch 2>/dev/null
ls $O | head -80
