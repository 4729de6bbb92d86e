# every kernel launch of one SVGP step, in start order (rocprofv3 kernel trace + profiles/timeline.py with threshold 0)
# usage: step_listing.sh <tag> [bench args]
cd /tmp; export TMPDIR=/tmp
tag=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/listing_$tag
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras "$@" > $O/log.txt 2>&1
python $GRAFT_REPO_ROOT/profiles/timeline.py $(find $O -name "*kernel_trace.csv") 0.0 | cut -c1-130 > $GRAFT_REPO_ROOT/gpurun_out/listing_$tag.txt
rm -rf $O
head -3 $GRAFT_REPO_ROOT/gpurun_out/listing_$tag.txt
