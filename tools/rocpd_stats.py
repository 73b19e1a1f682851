"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel stats CSV
(same columns as rocprofv3's kernel_stats.csv) plus a per-launch listing of the last N launches."""
import sqlite3
import sys


def main(db, out_csv):
    c = sqlite3.connect(db)
    rows = c.execute('select name, (end-start) from kernels').fetchall()
    agg = {}
    for n, d in rows:
        a = agg.setdefault(n, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    with open(out_csv, 'w') as f:
        f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
        for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write('"%s",%d,%d,%.1f,%.2f,%d,%d\n' % (n.replace('"', "'"), a[0], a[1], a[1] / a[0], 100.0 * a[1] / tot, a[2], a[3]))
    print('wrote', out_csv, 'kernels:', len(agg), 'total ms: %.3f' % (tot / 1e6))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
