#!/usr/bin/env python
"""Per-kernel sums of rocprofv3 --pmc counters from one or more rocpd sqlite databases -> CSV (for profiles/).

    python tools/pmc_summary.py out.csv run1_results.db [run2_results.db ...] [--note "text"]

Each database is one rocprofv3 pass (counters that share a pass sit in one db; FETCH_SIZE and WRITE_SIZE need separate
passes on gfx950).  Output: one row per kernel name: launches, then every counter summed over the launches, then derived
columns when their inputs are present:
  total_ms         = sum of the launches' durations (under the profiler)
  mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * total_ns * 2.4 GHz): matrix-pipe busy share against the
                     NOMINAL clock, i.e. directly comparable with "fraction of the 2.5 PF dense peak" (the chip clocks
                     below 2.4 GHz under MFMA load, so 1.0 is not reachable)
  wait_any_frac    = SQ_WAIT_ANY / SQ_WAVE_CYCLES, wait_inst_frac, active_frac likewise (quad-cycle units cancel)
  valu_per_mfma    = SQ_INSTS_VALU / SQ_INSTS_MFMA"""
import sqlite3
import sys


def main():
    args = sys.argv[1:]
    note = ''
    if '--note' in args:
        i = args.index('--note')
        note = args[i + 1]
        del args[i:i + 2]
    out, dbs = args[0], args[1:]
    data, launches, counters = {}, {}, []
    for db in dbs:
        c = sqlite3.connect(db)
        rows = c.execute('select kernel_name, counter_name, sum(value), count(*), sum(duration) from counters_collection group by kernel_name, counter_name').fetchall()
        for k, cn, v, n, dur in rows:
            data.setdefault(k, {})[cn] = v
            launches[k] = n
            if cn == 'SQ_VALU_MFMA_BUSY_CYCLES' or 'total_ns' not in data[k]:
                data[k]['total_ns'] = dur
            if cn not in counters:
                counters.append(cn)
    der = ['mfma_busy_frac', 'wait_any_frac', 'wait_inst_frac', 'active_frac', 'valu_per_mfma']
    with open(out, 'w') as f:
        if note:
            f.write('# ' + note + '\n')
        f.write('kernel,launches,total_ms,' + ','.join(counters + der) + '\n')
        for k in sorted(data, key=lambda k: -data[k].get('SQ_BUSY_CYCLES', data[k].get('FETCH_SIZE', 0))):
            d = data[k]
            g = d.get
            dv = []
            dv.append(g('SQ_VALU_MFMA_BUSY_CYCLES') / (1024.0 * g('total_ns') * 2.4) if g('SQ_VALU_MFMA_BUSY_CYCLES') and g('total_ns') else '')
            for nm in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY'):
                dv.append(g(nm) / g('SQ_WAVE_CYCLES') if g(nm) is not None and g('SQ_WAVE_CYCLES') else '')
            dv.append(g('SQ_INSTS_VALU') / g('SQ_INSTS_MFMA') if g('SQ_INSTS_VALU') and g('SQ_INSTS_MFMA') else '')
            f.write('"%s",%d,%.3f,' % (k.replace('"', "'"), launches[k], (g('total_ns') or 0) / 1e6) + ','.join('%.6g' % d[cn] if cn in d else '' for cn in counters)
                    + ',' + ','.join('%.4f' % v if v != '' else '' for v in dv) + '\n')
    print('wrote', out, 'kernels', len(data))


if __name__ == '__main__':
    main()
