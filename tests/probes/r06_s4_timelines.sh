#!/bin/bash
# fresh timelines of the exact-GP MAP step and the deep GP step (all launches >= 0.02 ms), for the r06 session-4 work
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/s4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_gp -o bench -- python $R/bench.py --workload gp --dtype float64 --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_trace_gp.log 2>&1
python $R/profiles/timeline.py $(find $O/trace_gp -name "*kernel_trace.csv") 0.02 > $O/gp_timeline.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_dgp -o bench -- python $R/bench.py --workload deepgp --samples 32 --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_trace_dgp.log 2>&1
python $R/profiles/timeline.py $(find $O/trace_dgp -name "*kernel_trace.csv") 0.03 > $O/dgp_timeline.txt 2>&1
rm -rf $O/trace_gp $O/trace_dgp
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > $O/tests_summary.txt
