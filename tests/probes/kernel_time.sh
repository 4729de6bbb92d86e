# average duration of the step's kernels matching a pattern, for a given build:  kernel_time.sh PATTERN [bench args]   (MXF_GP_LIB selects the build)
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/ktime
rm -rf $O
P=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras "$@" > /dev/null 2>&1
python - "$P" $(find $O -name '*kernel_stats.csv') <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if sys.argv[1] in r['Name']:
        print('%-70s calls %4s  avg %9.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
