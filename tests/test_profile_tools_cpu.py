"""CPU: the committed roofline tables are GENERATED from the committed per-launch profiles (tools/roofline_table.py — no typed numbers): regenerating
round 5's from `profiles/r05_bench_last_forward.csv` must reproduce the committed file, account for every launch of the forward (a ResBlock1 run as
two chain launches included) and add up to the workload's algorithmic FLOPs (1 168 559 FLOP per sample x 64 x 192 064 samples)."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.parametrize('rnd', ['r05', 'r06'])
def test_roofline_table_regenerates_from_the_committed_launch_list(tmp_path, capsys, rnd):
    from tools import roofline_table
    src = os.path.join(ROOT, 'profiles', '%s_bench_last_forward.csv' % rnd)
    dst = tmp_path / 'roofline.md'
    roofline_table.main(src, str(dst))
    capsys.readouterr()
    new = dst.read_text()
    old = open(os.path.join(ROOT, 'profiles', '%s_roofline.md' % rnd)).read()
    assert new == old
    n_launches = sum(1 for _ in open(src)) - 1
    assert '(%d launches of one forward' % n_launches in new
    m = re.search(r'whole forward: ([0-9.]+) ms of kernel time, ([0-9.]+) TFLOP', new)
    assert m and abs(float(m.group(2)) - 14.364) < 0.002           # 64 x 192 064 samples x 1 168 559 FLOP
    rows = {l.split('|')[1].strip(): int(l.split('|')[2]) for l in new.splitlines() if l.startswith('| stage')  and ' K=' in l}
    assert rows['stage 3 (C=64, L=48016) K=11'] == 2 and rows['stage 3 (C=64, L=48016) K=7'] == 2 and rows['stage 3 (C=64, L=48016) K=3'] == 1
