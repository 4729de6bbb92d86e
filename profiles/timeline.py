"""One training step as a timeline, from a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv): the dispatches between the last two Adam
launches, start / end / duration in ms relative to the step start, the hardware queue, launches >= min_ms only, plus per-kernel totals.
usage: timeline.py kernel_trace.csv [min_ms=0.15]"""
import csv
import re
import sys
from collections import defaultdict


def main(path, min_ms=0.15):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
    if len(adam) < 2:
        raise SystemExit('need at least two steps (adam_kernel launches) in the trace')
    step = rows[adam[-2] + 1:adam[-1] + 1]
    t0 = int(rows[adam[-2]]['End_Timestamp'])
    t1 = int(step[-1]['End_Timestamp'])
    print('# one step: %.2f ms, %d kernel launches (start end dur [ms] queue kernel; launches >= %.2f ms)' % ((t1 - t0) / 1e6, len(step), min_ms))
    tot = defaultdict(lambda: [0, 0.0])
    for r in step:
        s, e = (int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - t0) / 1e6
        name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        name = re.sub(r'^void ', '', name)
        key = name.split('(')[0][:70]
        tot[key][0] += 1
        tot[key][1] += e - s
        if e - s >= min_ms:
            print('%8.2f %8.2f %7.2f q%s %s' % (s, e, e - s, r['Queue_Id'], name[:110]))
    print('# per kernel: calls, total ms (sum of launch durations; concurrent launches overlap)')
    for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:25]:
        print('%6d %9.3f  %s' % (n, t, k))


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.15)
