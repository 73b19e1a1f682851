"""hifigan/streams.py::_reserve — the package's streams are taken from torch's pool in ONE go and in a fixed order (text stream, eight side streams, exchange
stream), whichever feature asks first: the HIP runtime maps streams onto its four hardware queues in creation order, and a training step ran 61.5 or 69 ms
depending on what had created streams before it (profiles/r06_branch_stream_queues.log).  No GPU needed: torch.cuda.Stream is replaced by a counter."""
import importlib

import pytest
import torch


class _FakeStream:
    made = []

    def __init__(self, device=None, priority=0):
        self.idx = len(_FakeStream.made)
        self.device = device
        self.priority = priority
        self.cuda_stream = 1000 + self.idx
        _FakeStream.made.append(self)


@pytest.fixture(autouse=True)
def _forget_fake_streams():
    yield
    from ttscube_amd.hifigan import streams as S
    importlib.reload(S)      # (a later test of the same session must not find the fake streams in the registry)


def _fresh(monkeypatch):
    from ttscube_amd.hifigan import streams as S
    S = importlib.reload(S)
    _FakeStream.made = []
    monkeypatch.setattr(torch.cuda, 'Stream', _FakeStream)
    monkeypatch.setattr(torch.cuda, 'current_device', lambda: 0)
    return S


@pytest.mark.parametrize('first', ['text', 'side', 'branch', 'exchange'])
def test_reserved_order_does_not_depend_on_the_first_user(monkeypatch, first):
    S = _fresh(monkeypatch)
    dev = torch.device('cuda', 0)
    {'text': lambda: S.text_stream(dev), 'side': lambda: S._side_streams(dev, 3), 'branch': lambda: S.branch_stream_handles(dev),
     'exchange': lambda: S.exchange_stream(dev)}[first]()
    assert len(_FakeStream.made) == 1 + S.N_RESERVED + 1           # everything was taken at the first touch ...
    assert S.text_stream(dev).idx == 0                              # ... text stream first,
    assert [s.idx for s in S._side_streams(dev, S.N_RESERVED)] == list(range(1, 1 + S.N_RESERVED))   # then the side streams,
    assert S.exchange_stream(dev).idx == 1 + S.N_RESERVED           # then the exchange stream
    assert S.branch_stream_handles(dev) == (1001, 1002)             # the generator's branch schedule borrows side streams 0 and 1
    assert len(_FakeStream.made) == 1 + S.N_RESERVED + 1           # and nothing new is created by asking again
    assert all(s.priority == 0 for s in _FakeStream.made)           # one priority: one hardware-queue pool


def test_more_side_streams_than_reserved_are_appended(monkeypatch):
    S = _fresh(monkeypatch)
    dev = torch.device('cuda', 0)
    ss = S._side_streams(dev, S.N_RESERVED + 2)
    assert [s.idx for s in ss[:S.N_RESERVED]] == list(range(1, 1 + S.N_RESERVED)) and len(ss) == S.N_RESERVED + 2
    assert S.side_streams_of(dev) == ss
    # a second device has a reservation of its own
    n = len(_FakeStream.made)
    S.text_stream(torch.device('cuda', 1))
    assert len(_FakeStream.made) == n + 1 + S.N_RESERVED + 1
