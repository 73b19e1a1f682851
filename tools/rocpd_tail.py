"""Launch-by-launch listing of the last MS milliseconds of a rocprofv3 rocpd database: start offset, duration, idle gap to the previous launch's end
(on the device, any stream), kernel name.     python tools/rocpd_tail.py run_results.db MS > listing.txt"""
import sqlite3
import sys

db, ms = sys.argv[1], float(sys.argv[2])
rows = sqlite3.connect(db).execute('select name, start, end from kernels order by start').fetchall()
t_end = max(r[2] for r in rows)
win = [r for r in rows if r[1] >= t_end - ms * 1e6]
t0 = win[0][1]
prev_end = t0
busy = 0
print('%9s %9s %8s  %s' % ('start us', 'dur us', 'gap us', 'kernel'))
for n, s, e in win:
    gap = (s - prev_end) / 1e3
    print('%9.1f %9.1f %8.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, n[:110]))
    prev_end = max(prev_end, e)
    busy += e - s
print('window %.1f us, %d launches, sum of durations %.1f us' % ((t_end - t0) / 1e3, len(win), busy / 1e3))
