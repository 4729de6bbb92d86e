# one exact-GP float64 step (configs[1]) as a kernel timeline
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/gptl
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o b -- python $GRAFT_REPO_ROOT/bench.py --workload gp --dtype float64 --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/log.txt 2>&1
python $GRAFT_REPO_ROOT/profiles/timeline.py $(find $O -name "*kernel_trace.csv") 0.1 > $O/timeline.txt 2>&1
head -n 90 $O/timeline.txt | cut -c1-170
