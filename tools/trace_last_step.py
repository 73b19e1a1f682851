"""Aggregate, per kernel name, the kernels of a rocprofv3 kernel trace (csv) that started in the final `window_ms`
milliseconds of the timeline (= the last, warm step):  python tools/trace_last_step.py trace.csv window_ms out.csv"""
import collections
import csv
import sys


def main(path, window_ms, out):
    rows = list(csv.DictReader(open(path)))
    t_end = max(int(r['End_Timestamp']) for r in rows)
    win = [r for r in rows if int(r['Start_Timestamp']) > t_end - int(window_ms * 1e6)]
    agg = collections.defaultdict(lambda: [0, 0])
    for r in win:
        a = agg[r['Kernel_Name']]
        a[0] += 1
        a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    tot = sum(a[1] for a in agg.values())
    with open(out, 'w') as f:
        f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage"\n')
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write('"%s",%d,%d,%.1f,%.2f\n' % (k[:200].replace('"', "'"), a[0], a[1], a[1] / a[0], 100.0 * a[1] / tot))
    print('window kernels: %d, GPU time %.2f ms' % (len(win), tot / 1e6))


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]), sys.argv[3])
