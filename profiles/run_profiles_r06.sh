#!/bin/bash
# Round-6 profiles (run on the GPU box through gpurun, from the repo root); the summaries are copied into profiles/ afterwards.
# Counters are collected in their own passes with --kernel-trace only (never with sys / hip / hsa traces).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. rocprofv3 kernel stats of the default bench command (the roofline kernels' per-launch averages) and of the bare step; one step as a timeline
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_full -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_trace_full.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_step -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_trace_step.log 2>&1
python $R/profiles/timeline.py $(find $O/trace_step -name "*kernel_trace.csv") 0.15 > $O/r06_step_timeline.txt 2>&1
# 2. the whitened float32 form at trained-like parameters
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_whitened -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --trained-like > $O/bench_trace_whitened.log 2>&1
python $R/profiles/timeline.py $(find $O/trace_whitened -name "*kernel_trace.csv") 0.15 > $O/r06_step_timeline_whitened.txt 2>&1
# 3. the per-rank share of an 8-GPU run (4 samples)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_s4 -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --samples 4 > $O/bench_trace_s4.log 2>&1
python $R/profiles/timeline.py $(find $O/trace_s4 -name "*kernel_trace.csv") 0.03 > $O/r06_step_timeline_4samples.txt 2>&1
# 4. the exact-GP MAP step (BASELINE configs[1])
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_gp -o bench -- python $R/bench.py --workload gp --dtype float64 --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_trace_gp.log 2>&1
python $R/profiles/timeline.py $(find $O/trace_gp -name "*kernel_trace.csv") 0.1 > $O/r06_exact_gp_timeline.txt 2>&1
# 4b. the deep GP step (BASELINE configs[4])
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_dgp -o bench -- python $R/bench.py --workload deepgp --samples 32 --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_trace_dgp.log 2>&1
python $R/profiles/timeline.py $(find $O/trace_dgp -name "*kernel_trace.csv") 0.25 > $O/r06_deepgp_timeline.txt 2>&1
for d in full step whitened s4 gp; do cp $(find $O/trace_$d -name "*kernel_stats.csv" | head -1) $O/r06_kernel_stats_$d.csv; done
# 5. PMC passes: the Gram kernel (HBM traffic per launch), the T product of r05 (t), the K-major T product of r06 with its U row (tbt), Psi2
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_gram_write -o gram -- python $R/tests/probes/gram_f32.py > $O/pmc_gram_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_gram_fetch -o gram -- python $R/tests/probes/gram_f32.py > $O/pmc_gram_fetch.log 2>&1
python $R/profiles/pmc_summary.py gram_lean_kernel $O/r06_gram_pmc.json $O/pmc_gram_write $O/pmc_gram_fetch > $O/r06_gram_pmc.txt 2>&1
for w in t tbt psi2; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_${w}_fetch -o g -- python $R/tests/probes/split_pmc.py $w > $O/pmc_${w}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_${w}_write -o g -- python $R/tests/probes/split_pmc.py $w > $O/pmc_${w}_write.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${w}_sq -o g -- python $R/tests/probes/split_pmc.py $w > $O/pmc_${w}_sq.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_${w}_lds -o g -- python $R/tests/probes/split_pmc.py $w > $O/pmc_${w}_lds.log 2>&1
  python $R/profiles/pmc_summary.py gemm_f16x2 $O/r06_gemm_${w}_pmc.json $O/pmc_${w}_fetch $O/pmc_${w}_write $O/pmc_${w}_sq $O/pmc_${w}_lds > $O/r06_gemm_${w}_pmc.txt 2>&1
done
rm -rf $O/trace_* $O/pmc_*_fetch $O/pmc_*_write $O/pmc_*_sq $O/pmc_*_lds $O/pmc_gram_write $O/pmc_gram_fetch 2>/dev/null
# 6. bench lines
cd $R
python bench.py > $O/r06_bench_full.json 2> $O/full.log
python bench.py --samples 4 --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/r06_bench_4samples_per_gpu.json 2>/dev/null
python bench.py --samples 4 --trained-like --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/r06_bench_4samples_per_gpu_trained_like.json 2>/dev/null
python bench.py --samples 4 --steps 30 --warmup 5 --no-cpu-baseline --no-extras --force-dist > $O/r06_bench_4samples_per_gpu_rccl_path.json 2>/dev/null
python bench.py --minibatch 8192 --steps 40 --warmup 5 --no-cpu-baseline > $O/r06_bench_minibatch8192.json 2>/dev/null
python bench.py --minibatch 8192 --samples 4 --steps 40 --warmup 5 --no-cpu-baseline > $O/r06_bench_minibatch8192_4samples.json 2>/dev/null
python bench.py --minibatch 8192 --shard rows --samples 1 --steps 40 --warmup 5 --no-cpu-baseline > $O/r06_bench_rows_minibatch8192_1gpu.json 2>/dev/null
python bench.py --minibatch 8192 --shard rows --samples 1 --proxy-world 8 --steps 40 --warmup 5 --no-cpu-baseline > $O/r06_bench_rows_minibatch8192_rank_of_8.json 2>/dev/null
python bench.py --workload gp --dtype float64 --steps 5 --warmup 2 --no-cpu-baseline > $O/r06_bench_exact_gp_f64.json 2>/dev/null
python bench.py --workload deepgp --samples 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/r06_bench_deepgp_4samples.json 2>/dev/null
python bench.py --workload deepgp --samples 32 --steps 5 --warmup 2 --no-cpu-baseline > $O/r06_bench_deepgp_32samples.json 2>/dev/null
# the distributed product with REAL kernels in 2 / 4 processes on this one GPU (gloo on device tensors): a software check, not a measurement
python bench.py --gpus 2 --backend gloo --same-device --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/r06_bench_2ranks_one_gpu_gloo.json 2>/dev/null
python bench.py --gpus 4 --backend gloo --same-device --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/r06_bench_4ranks_one_gpu_gloo.json 2>/dev/null
python bench.py --gpus 2 --backend gloo --same-device --minibatch 8192 --steps 10 --warmup 2 --no-cpu-baseline > $O/r06_bench_2ranks_one_gpu_gloo_minibatch8192.json 2>/dev/null
for f in $O/*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],3), d.get('ms_per_step_trained_like'), d.get('roofline',{}).get('frac'), d.get('roofline_mfma',{}).get('frac'), d.get('roofline_mfma_psi2',{}).get('frac'), d.get('roofline_f64_mfma',{}).get('frac'), d.get('ranks'), d.get('collectives_per_step'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
python tests/probes/svgp_stages.py 32 > $O/r06_stage_stamps.txt 2>&1
python tests/probes/svgp_stages.py 4 > $O/r06_stage_stamps_4samples.txt 2>&1
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" > $O/r06_gpu_suite_summary.txt
ls $O
