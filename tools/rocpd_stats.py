"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel stats CSV (same columns as rocprofv3's
kernel_stats.csv) and, optionally, a per-launch listing of the last N launches (one forward / one step, in launch order).

    python tools/rocpd_stats.py run_results.db stats.csv [N last.csv]
    python tools/rocpd_stats.py run_results.db stats.csv -MS window_stats.csv     # per-kernel stats of the last MS milliseconds
    python tools/rocpd_stats.py run_results.db stats.csv fwd:PREFIX last.csv      # launches of the LAST forward: from the last kernel
                                                                                  # whose name contains PREFIX back to (not including)
                                                                                  # the previous launch of the forward's final kernel"""
import sqlite3
import sys


def main(db, out_csv, n_last=0, last_csv=None):
    c = sqlite3.connect(db)
    rows = c.execute('select name, start, end, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count from kernels order by start').fetchall()
    agg = {}
    for r in rows:
        n, d = r[0], r[2] - r[1]
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
    if isinstance(n_last, str) and last_csv:
        # one forward = the run of launches that ends with the LAST launch of its final kernel (conv_cout1_kernel = conv_post) and
        # starts right after the previous launch of that kernel; only kernels whose name contains the prefix count (torch's own
        # launches between two forwards — the bench's isfinite / copies — are skipped), so the file holds exactly one forward
        # the forward's final kernel: conv_post (conv_cout1_kernel), or — when conv_post rides in the epilogue of the last chain launch (round 4) —
        # the chain instantiation with the POST flag (last template argument true after the interleave flag)
        pref = n_last.split(':', 1)[1]
        is_final = lambda name: 'conv_cout1_kernel' in name or ('rbchain_f16x3_kernel' in name and name.split('>')[0].rstrip().endswith('true, true'))
        mine = [r for r in rows if pref in r[0]]
        ends = [i for i, r in enumerate(mine) if is_final(r[0])]
        assert len(ends) >= 2, 'need at least two forwards in the trace'
        fwd = mine[ends[-2] + 1:ends[-1] + 1]
        with open(last_csv, 'w') as f:
            f.write('"Name","DurationNs","GridX","GridY","GridZ","WorkgroupX","LdsBytes","Vgprs"\n')
            for r in fwd:
                f.write('"%s",%d,%d,%d,%d,%d,%d,%d\n' % (r[0].replace('"', "'"), r[2] - r[1], r[3], r[4], r[5], r[6], r[7], r[8]))
        print('wrote', last_csv, '%d launches of one forward, sum ms: %.3f' % (len(fwd), sum(r[2] - r[1] for r in fwd) / 1e6))
    elif n_last < 0 and last_csv:   # steady-state window: everything that started in the last |n_last| ms of the trace
        t_end = max(r[2] for r in rows)
        win = [r for r in rows if r[1] >= t_end + n_last * 1e6]
        agg = {}
        for r in win:
            a = agg.setdefault(r[0], [0, 0])
            a[0] += 1
            a[1] += r[2] - r[1]
        tot = sum(a[1] for a in agg.values())
        with open(last_csv, 'w') as f:
            f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage"\n')
            for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                f.write('"%s",%d,%d,%.1f,%.2f\n' % (n.replace('"', "'"), a[0], a[1], a[1] / a[0], 100.0 * a[1] / tot))
        print('wrote', last_csv, 'window %d ms: %d launches, busy %.3f ms' % (-n_last, len(win), tot / 1e6))
    elif n_last and last_csv:
        with open(last_csv, 'w') as f:
            f.write('"Name","DurationNs","GridX","GridY","GridZ","WorkgroupX","LdsBytes","Vgprs"\n')
            for r in rows[-n_last:]:
                f.write('"%s",%d,%d,%d,%d,%d,%d,%d\n' % (r[0].replace('"', "'"), r[2] - r[1], r[3], r[4], r[5], r[6], r[7], r[8]))
        print('wrote', last_csv, 'sum ms: %.3f' % (sum(r[2] - r[1] for r in rows[-n_last:]) / 1e6))


if __name__ == '__main__':
    a = sys.argv
    n = (a[3] if a[3].startswith('fwd:') else int(a[3])) if len(a) > 3 else 0
    main(a[1], a[2], n, a[4] if len(a) > 4 else None)
