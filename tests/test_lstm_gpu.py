"""GPU parity (through the C ABI): Linear (MFMA NT GEMM) and LSTM/BiLSTM kernels vs torch fp32 CPU."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('M,N,K,act', [(300, 2048, 256, None), (77, 80, 128, None), (1000, 101, 512, None), (5, 2, 512, 'sigmoid'),
                                        (129, 256, 641, None), (64, 256, 80, 'relu'), (3, 240, 512, None), (260, 512, 1280, 'tanh')])
def test_linear_matches_torch(M, N, K, act):
    from ttscube_amd.hip_layers import linear_hip
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    ref = torch.nn.functional.linear(x, w, b)
    ref = {'sigmoid': torch.sigmoid, 'relu': torch.relu, 'tanh': torch.tanh, None: lambda v: v}[act](ref)
    y = linear_hip(x.cuda(), w.cuda(), b.cuda(), act=act).cpu()
    assert y.shape == ref.shape
    assert float((y - ref).abs().max()) < 2e-5


@pytest.mark.parametrize('inp,H,layers,bidir,B,T', [(64, 256, 2, True, 2, 37), (640, 256, 2, True, 1, 150), (641, 64, 2, True, 3, 90),
                                                     (1280, 512, 2, False, 2, 12), (640, 512, 2, True, 1, 25), (300, 256, 1, True, 2, 5)])
def test_lstm_matches_torch(inp, H, layers, bidir, B, T):
    from ttscube_amd.hip_layers import LSTMHip
    torch.manual_seed(inp + H + T)
    m = nn.LSTM(input_size=inp, hidden_size=H, num_layers=layers, bidirectional=bidir, batch_first=True)
    x = torch.randn(B, T, inp)
    with torch.no_grad():
        ref, (hr, cr) = m(x)
    m = m.cuda()
    y, (hn, cn) = LSTMHip(m)(x.cuda(), return_state=True)
    assert float((y.cpu() - ref).abs().max()) < 2e-5
    assert float((hn.cpu() - hr).abs().max()) < 2e-5 and float((cn.cpu() - cr).abs().max()) < 5e-5


def test_lstm_ragged_batch_equals_per_utterance():
    """pack_padded_sequence semantics: a padded batch reproduces each utterance run alone (SURVEY §7 'ragged batches')."""
    from ttscube_amd.hip_layers import LSTMHip
    torch.manual_seed(5)
    m = nn.LSTM(input_size=96, hidden_size=128, num_layers=2, bidirectional=True, batch_first=True).cuda()
    h = LSTMHip(m)
    lens = [17, 5, 30, 1]
    x = torch.randn(4, 30, 96).cuda()
    y = h(x, lengths=lens)
    for b, n in enumerate(lens):
        solo = h(x[b:b + 1, :n])
        assert torch.equal(y[b, :n], solo[0])
        assert bool((y[b, n:] == 0).all())


def test_lstm_initial_state_chaining():
    """running T steps at once == running them in two calls that pass (h, c) along (needed by the AR mel decoder)."""
    from ttscube_amd.hip_layers import LSTMHip
    torch.manual_seed(6)
    m = nn.LSTM(input_size=40, hidden_size=64, num_layers=2, bidirectional=False, batch_first=True).cuda()
    h = LSTMHip(m)
    x = torch.randn(2, 9, 40).cuda()
    y, st = h(x, return_state=True)
    y1, st1 = h(x[:, :4], return_state=True)
    y2, st2 = h(x[:, 4:], hx=st1, return_state=True)
    assert torch.equal(torch.cat([y1, y2], 1), y) and torch.equal(st2[0], st[0]) and torch.equal(st2[1], st[1])


def test_lstm_two_utterances_per_workgroup():
    """lstm_seq_kernel<2> (B * ndir > 512 sequences: two utterances share one W_hh stream), ragged, vs torch.nn.LSTM."""
    from ttscube_amd.hip_layers import LSTMHip
    torch.manual_seed(9)
    m = nn.LSTM(input_size=24, hidden_size=96, num_layers=1, bidirectional=True, batch_first=True)   # (H = 64 / 128 run the resident kernel)
    B, T = 301, 6
    x = torch.randn(B, T, 24)
    lens = [(b % T) + 1 for b in range(B)]
    with torch.no_grad():
        packed = nn.utils.rnn.pack_padded_sequence(x, lens, batch_first=True, enforce_sorted=False)
        ref, _ = nn.utils.rnn.pad_packed_sequence(m(packed)[0], batch_first=True, total_length=T)
    y = LSTMHip(m.cuda())(x.cuda(), lengths=lens).cpu()
    assert float((y - ref).abs().max()) < 2e-5


@pytest.mark.parametrize('H,B,T', [(256, 40, 23), (256, 3, 50), (512, 3, 21), (512, 10, 9), (64, 70, 33), (128, 5, 40), (256, 65, 30), (256, 133, 17), (512, 33, 7)])
def test_lstm_resident_kernels_ragged_vs_torch_and_solo(H, B, T):
    """Register-resident recurrences: lstm_seq_resident_kernel (H = 64 / 128), lstm_seq_split_res_kernel with 4 (H = 256) / 16
    (H = 512) members.  More (utterance, direction) pairs than one launch holds (256 CUs / members): 2 or 4 utterances per member group
    (lstm_seq_split_res_nb_kernel) — B = 40 / 65 at H = 256 and B = 10 at H = 512 run two per group (65: a ragged last group),
    B = 133 at H = 256 and B = 33 at H = 512 four per group in two launches.  Ragged batch against torch.nn.LSTM (packed), and
    bit-identical to each utterance run alone (the one-utterance kernel)."""
    from ttscube_amd.hip_layers import LSTMHip
    torch.manual_seed(H + B + T)
    m = nn.LSTM(input_size=48, hidden_size=H, num_layers=2, bidirectional=True, batch_first=True)
    x = torch.randn(B, T, 48)
    lens = [((7 * b) % T) + 1 for b in range(B)]
    lens[0] = T
    with torch.no_grad():
        packed = nn.utils.rnn.pack_padded_sequence(x, lens, batch_first=True, enforce_sorted=False)
        ref, _ = nn.utils.rnn.pad_packed_sequence(m(packed)[0], batch_first=True, total_length=T)
    h = LSTMHip(m.cuda())
    y = h(x.cuda(), lengths=lens)
    assert float((y.cpu() - ref).abs().max()) < 3e-5
    for b in sorted(v for v in {0, 1, 2, 3, B // 2, B - 2, B - 1} if 0 <= v < B):
        solo = h(x[b:b + 1, :lens[b]].cuda())
        assert torch.equal(y[b, :lens[b]], solo[0])
        assert bool((y[b, lens[b]:] == 0).all())


def test_two_handles_on_two_streams_do_not_share_handoff_state():
    """the split recurrences keep their counters / granule rings / abort words per (device, stream)
    (csrc/util.cpp handoff_area), so two handles driven concurrently on two streams of one device cannot corrupt each other."""
    from ttscube_amd import _lib
    from ttscube_amd.hip_layers import LSTMHip
    torch.manual_seed(11)
    ma = nn.LSTM(input_size=96, hidden_size=256, num_layers=2, bidirectional=True, batch_first=True).cuda()
    mb = nn.LSTM(input_size=80, hidden_size=512, num_layers=1, bidirectional=True, batch_first=True).cuda()
    ha, hb = LSTMHip(ma), LSTMHip(mb)
    xa, xb = torch.randn(2, 140, 96).cuda(), torch.randn(1, 90, 80).cuda()
    ra, rb = ha(xa), hb(xb)          # solo, default stream
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for _ in range(6):
        with torch.cuda.stream(sa):
            ya = ha(xa)
        with torch.cuda.stream(sb):
            yb = hb(xb)
        outs.append((ya, yb))
    torch.cuda.synchronize()
    assert _lib.lib().ttsc_lstm_split_status() == 0
    for ya, yb in outs:
        assert torch.equal(ya, ra) and torch.equal(yb, rb)


@pytest.mark.parametrize('H,B,T,n', [(256, 21, 19, 8), (256, 21, 19, 4), (256, 6, 25, 2), (512, 9, 8, 8), (256, 300, 5, 0)])
def test_lstm_group_size_does_not_change_results(H, B, T, n):
    """ttsc_lstm_set_group_size: n utterances per member group (8 / 4 / 2, ragged last group; B = 300 at H = 256: automatic -> 4 per group,
    three launches... and 8 once that is not enough) — bit-identical to one utterance per group, forward and saved training state."""
    from ttscube_amd import _lib
    from ttscube_amd.hip_layers import LSTMHip
    torch.manual_seed(H + B + n)
    m = nn.LSTM(input_size=32, hidden_size=H, num_layers=1, bidirectional=True, batch_first=True).cuda()
    h = LSTMHip(m)
    x = torch.randn(B, T, 32).cuda()
    lens = [((5 * b) % T) + 1 for b in range(B)]
    with _lib.lstm_group_size(n):
        y = h(x, lengths=lens)
    if n == 0:
        idx = list(range(0, B, 37)) + [B - 1]
        for b in idx:
            solo = h(x[b:b + 1, :lens[b]])
            assert torch.equal(y[b, :lens[b]], solo[0])
    else:
        with _lib.lstm_group_size(1):
            y1 = h(x, lengths=lens)
        assert torch.equal(y, y1)
    assert int(_lib.lib().ttsc_lstm_set_group_size(3)) == -1
    _lib.check_split_status('test')
