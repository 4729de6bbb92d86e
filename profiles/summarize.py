"""Turns a rocprofv3 (--kernel-trace --stats, rocpd sqlite output) database into the text summary committed here."""
import re
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"))
    with open(out, 'w') as f:
        f.write('# rocprofv3 --kernel-trace --stats summary (durations in microseconds)\n')
        f.write('%8s %14s %14s %7s  %s\n' % ('calls', 'total_us', 'avg_us', 'pct', 'kernel'))
        for name, calls, tot, avg, pct in rows:
            name = re.sub(r'\(anonymous namespace\)::', '', name)
            f.write('%8d %14.1f %14.1f %6.2f%%  %s\n' % (calls, tot, avg, pct, name[:160]))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
