"""CPU: `python bench.py --gpus N` without a launcher starts its own N ranks (the driver's N = 1 command is a plain `python bench.py`; if
its N > 1 command has the same shape the run must not die on the WORLD_SIZE assertion) — one process per GPU through torch.distributed.run, rendezvous
on 127.0.0.1, every bench flag handed through; under a launcher (WORLD_SIZE set) nothing is spawned."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _run_main(monkeypatch, argv, env_world=None):
    import bench
    calls = []

    class Done:
        returncode = 0

    def fake_run(cmd, env=None, **kw):
        calls.append((list(cmd), dict(env or {})))
        return Done()

    monkeypatch.setattr(subprocess, 'run', fake_run)
    monkeypatch.setattr(sys, 'argv', ['bench.py'] + argv)
    if env_world is None:
        monkeypatch.delenv('WORLD_SIZE', raising=False)
    else:
        monkeypatch.setenv('WORLD_SIZE', str(env_world))
    return bench, calls


@pytest.mark.parametrize('mode', [None, 'train', 'e2e'])
def test_gpus_n_without_a_launcher_spawns_n_ranks(monkeypatch, mode):
    argv = ['--gpus', '4', '--steps', '3', '--warmup', '1'] + (['--mode', mode] if mode else [])
    bench, calls = _run_main(monkeypatch, argv)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '4' and '--nnodes=1' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and int(cmd[cmd.index('--master-port') + 1]) > 0
    script = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[script + 1:] == argv                      # every flag of the call reaches the ranks
    assert env.get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'  # dmabuf IPC: RCCL across processes needs it on this driver


def test_under_a_launcher_nothing_is_spawned(monkeypatch):
    """WORLD_SIZE set = a launcher started this rank: the spawn branch is skipped (and the bench proper then refuses to run without a GPU)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip('on a GPU box the bench proper would start a 2-rank rendezvous')
    bench, calls = _run_main(monkeypatch, ['--gpus', '2', '--steps', '1', '--warmup', '0'], env_world=2)
    with pytest.raises((AssertionError, RuntimeError, SystemExit, Exception)):
        bench.main()
    assert calls == []


def test_live_traffic_reads_the_counter_databases(monkeypatch, tmp_path):
    """bench.py::live_traffic — the three rocprofv3 --pmc child passes are replaced by a fake that leaves the rocpd database such a pass leaves: the bytes of ONE
    forward are (2 x FETCH_SIZE + WRITE_SIZE) KiB over the ttsc:: kernels / 3 forwards, the register-only probe loop and foreign kernels do not count, the
    write calibration compares the waveform-writing launch with B x L x 4 bytes; a failing pass returns None (the caller keeps the committed summary)."""
    import sqlite3
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench
    B, T = 2, 10
    L = T
    for u, k in ((5, 16), (3, 16), (4, 4), (4, 4)):
        L = (L - 1) * u - 2 * ((k - u) // 2) + k
    post = 'void ttsc::rbchain_f16x3_kernel<1, 11, 4, 8, 2, 6, 2, 1, true, true>(ttsc::ChainArgs)'
    values = {'FETCH_SIZE': [('void ttsc::conv_f16x3_wide_kernel<256, 3, 1>(ttsc::ConvArgs)', 3000.0, 3), (post, 600.0, 3), ('at::native::copy', 999.0, 3)],
              'WRITE_SIZE': [('void ttsc::conv_f16x3_wide_kernel<256, 3, 1>(ttsc::ConvArgs)', 1500.0, 3), (post, 3 * B * L * 4 / 1024.0, 3)],
              'SQ_INSTS_MFMA': [('void ttsc::conv_f16x3_wide_kernel<256, 3, 1>(ttsc::ConvArgs)', 9000.0, 3), ('ttsc::mfma_sustained_kernel(int)', 1e9, 3)]}
    fail = {'on': None}

    class FakePopen:
        def __init__(self, cmd, **kw):
            cn, d = cmd[cmd.index('--pmc') + 1], cmd[cmd.index('-d') + 1]
            assert '--kernel-trace' in cmd and '--no-extra' in cmd and kw.get('start_new_session')
            self.pid, self.returncode = 12345, (1 if fail['on'] == cn else 0)
            if self.returncode == 0:
                os.makedirs(os.path.join(d, 'host'), exist_ok=True)
                c = sqlite3.connect(os.path.join(d, 'host', 'p_results.db'))
                c.execute('create table counters_collection (kernel_name text, counter_name text, value real, duration real)')
                for k, v, n in values[cn]:
                    for _ in range(n):
                        c.execute('insert into counters_collection values (?, ?, ?, 1.0)', (k, cn, v / n))
                c.commit()
                c.close()

        def wait(self, timeout=None):
            return self.returncode

    monkeypatch.setattr(subprocess, 'Popen', FakePopen)
    import shutil
    monkeypatch.setattr(shutil, 'which', lambda name: sys.executable)     # (any existing path: the fake never runs it)
    got, detail = bench.live_traffic('f16x3', B, T)
    assert got is not None, detail
    f, w = (3000.0 + 600.0) * 1024 / 3, (1500.0 + 3 * B * L * 4 / 1024.0) * 1024 / 3
    assert abs(got - (2 * f + w)) < 1e-6 * got
    assert abs(detail['write_calibration_ratio'] - 1.0) < 1e-9 and detail['sq_insts_mfma_per_forward'] == 3000.0
    fail['on'] = 'WRITE_SIZE'
    got, why = bench.live_traffic('f16x3', B, T)
    assert got is None and 'WRITE_SIZE' in why
