#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/s4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_dgp -o bench -- python $R/bench.py --workload deepgp --samples 4 --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_trace_dgp.log 2>&1
python $R/profiles/timeline.py $(find $O/trace_dgp -name "*kernel_trace.csv") 0.02 > $O/dgp4_timeline.txt 2>&1
rm -rf $O/trace_dgp
