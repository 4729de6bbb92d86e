#!/bin/bash
# Reproduces the profiles committed under profiles/ (run on the GPU box through gpurun, from the repo root).
# Counters are collected in their own passes with --kernel-trace only (never with sys/hip/hsa traces).
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_trace.log 2>&1
# the default bench command WITH its roofline micro-measurements (gram_kernel<float,8,0> at N=65536, gemm_split at the T shape): the
# per-kernel averages of this summary are what bench.py's roofline / roofline_mfma report from HIP events
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_full -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_trace_full.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o gram -- python $R/tests/probes/gram_f32.py > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o gram -- python $R/tests/probes/gram_f32.py > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o gram -- python $R/tests/probes/gram_f32.py > $O/pmc_sq.log 2>&1
ls -R $O | head -40
