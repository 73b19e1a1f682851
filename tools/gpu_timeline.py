"""Timeline view of a rocprofv3 rocpd kernel trace: how busy the GPU was over the last MS milliseconds of the run, how many kernels were in
flight at a time, where the idle gaps are and which kernels hold the critical stretches.

    python tools/gpu_timeline.py run_results.db MS [bucket_ms] > timeline.txt"""
import sqlite3
import sys


def short(n):
    n = n.replace('void ', '').replace('ttsc::', '')
    return n[:70]


def main(db, ms, bucket_ms=2.0):
    c = sqlite3.connect(db)
    rows = c.execute('select name, start, end from kernels order by start').fetchall()
    t_end = max(r[2] for r in rows)
    t0 = t_end - ms * 1e6
    rows = [(n, max(s, t0), e) for n, s, e in rows if e > t0]
    # sweep: concurrency level over time
    ev = []
    for n, s, e in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    level, last, hist = 0, t0, {}
    gaps = []
    for t, d in ev:
        if t > last:
            hist[level] = hist.get(level, 0) + (t - last)
            if level == 0 and t - last > 20e3:
                gaps.append((t - last, last))
        level += d
        last = t
    tot = t_end - t0
    print('window %.1f ms, %d launches, sum of kernel durations %.1f ms' % (tot / 1e6, len(rows), sum(e - s for _, s, e in rows) / 1e6))
    print('GPU idle (no kernel in flight): %.2f ms (%.1f %%)' % (hist.get(0, 0) / 1e6, 100.0 * hist.get(0, 0) / tot))
    for k in sorted(hist):
        if k:
            print('  %2d kernel(s) in flight: %7.2f ms' % (k, hist[k] / 1e6))
    print('idle gaps > 20 us: %d, total %.2f ms; largest:' % (len(gaps), sum(g for g, _ in gaps) / 1e6))
    ends = sorted(rows, key=lambda r: r[2])
    for g, at in sorted(gaps, reverse=True)[:12]:
        before = [r for r in ends if abs(r[2] - at) < 2]
        after = [r for r in rows if abs(r[1] - (at + g)) < 2]
        print('  %7.1f us at t = %6.2f ms   after %s   before %s' % (g / 1e3, (at - t0) / 1e6, short(before[0][0]) if before else '?', short(after[0][0]) if after else '?'))
    nb = int(tot / (bucket_ms * 1e6)) + 1
    print('per %.1f ms bucket: busy fraction (union), mean kernels in flight, the kernel with most time in the bucket' % bucket_ms)
    for b in range(nb):
        bs, be = t0 + b * bucket_ms * 1e6, min(t0 + (b + 1) * bucket_ms * 1e6, t_end)
        if be <= bs:
            break
        inb = [(n, max(s, bs), min(e, be)) for n, s, e in rows if e > bs and s < be]
        ivs = sorted((s, e) for _, s, e in inb)
        busy, cur_s, cur_e = 0, None, None
        for s, e in ivs:
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        if cur_e is not None:
            busy += cur_e - cur_s
        per = {}
        for n, s, e in inb:
            per[n] = per.get(n, 0) + e - s
        top = max(per.items(), key=lambda kv: kv[1]) if per else ('-', 0)
        print('  %6.1f ms  busy %.2f  in flight %.2f  launches %4d  %s (%.2f ms)' % ((bs - t0) / 1e6, busy / (be - bs), sum(e - s for _, s, e in inb) / (be - bs), len(inb),
                                                                               short(top[0]), top[1] / 1e6))


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]), float(sys.argv[3]) if len(sys.argv) > 3 else 2.0)
