#!/bin/bash
# rocprofv3 --kernel-trace --stats summaries of the bench commands (r05; the per-kernel averages the rooflines are checked against)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/stats_r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/full -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/full.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/step.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/whitened -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --trained-like > $O/whitened.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s4 -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras --samples 4 > $O/s4.log 2>&1
for d in full step whitened s4; do cp $(find $O/$d -name "*kernel_stats.csv" | head -1) $O/r05_kernel_stats_$d.csv; rm -rf $O/$d; done
cd $R
python bench.py --minibatch 8192 --shard rows --samples 1 --proxy-world 8 --steps 60 --warmup 5 --no-cpu-baseline > $O/r05_bench_rows_minibatch8192_rank_of_8.json 2>/dev/null
python bench.py --minibatch 65536 --shard rows --samples 1 --proxy-world 8 --steps 60 --warmup 5 --no-cpu-baseline > $O/r05_bench_rows_fullbatch_rank_of_8.json 2>/dev/null
ls $O
