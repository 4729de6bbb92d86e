#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/s4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_gp -o bench -- python $R/bench.py --workload gp --dtype float64 --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_trace_gp.log 2>&1
python $R/profiles/timeline.py $(find $O/trace_gp -name "*kernel_trace.csv") 0.05 > $O/gp_timeline2.txt 2>&1
rm -rf $O/trace_gp
