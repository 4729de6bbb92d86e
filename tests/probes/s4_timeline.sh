# one SVGP step at 4 samples (the per-rank share of an 8-GPU run) as a kernel timeline
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s4tl
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o b -- python $GRAFT_REPO_ROOT/bench.py --samples ${1:-4} --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/log.txt 2>&1
python $GRAFT_REPO_ROOT/profiles/timeline.py $(find $O -name "*kernel_trace.csv") 0.03 > $O/timeline.txt 2>&1
head -n 60 $O/timeline.txt | cut -c1-150
