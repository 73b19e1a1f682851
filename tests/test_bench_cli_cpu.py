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
