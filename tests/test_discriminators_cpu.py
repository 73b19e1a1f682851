"""CPU: (i) the batching identity the discriminator step relies on (hifigan/disc_hip.py::_pair: a weight-normed sub-discriminator sees real and
generated audio as ONE batch — outputs, feature maps, loss and parameter gradients equal two separate calls), shown on the torch-op formulation of
the same modules (tests/torch_reference.py); (ii) the drop-in classes of hifigan/discriminators.py: the reference's state_dict layout, and no CPU path."""
import pytest
import torch

from tests import torch_reference as TR
from ttscube_amd.hifigan import discriminators as D


def _grads(mod):
    return [p.grad.clone() for p in mod.parameters() if p.grad is not None]


def test_batched_pair_equals_two_calls():
    torch.manual_seed(0)
    y = torch.randn(2, 1, 1200)
    y_hat = torch.randn(2, 1, 1200)
    for d in (D.DiscriminatorP(3), D.DiscriminatorS()):
        outs = []
        for batched in (False, True):
            d.zero_grad(set_to_none=True)
            r, fr, g, fg = TR.disc_pair(d, y, y_hat, batched)
            loss, _, _ = TR.discriminator_loss([r], [g])
            loss = loss + sum(f.abs().mean() for f in fr) + sum(f.abs().mean() for f in fg)
            loss.backward()
            outs.append((r.detach(), g.detach(), [f.detach() for f in fr + fg], float(loss), _grads(d)))
        a, b = outs
        assert torch.allclose(a[0], b[0], atol=1e-5) and torch.allclose(a[1], b[1], atol=1e-5)
        assert all(torch.allclose(u, v, atol=1e-5) for u, v in zip(a[2], b[2]))
        assert abs(a[3] - b[3]) < 1e-5 * max(1.0, abs(a[3]))
        assert len(a[4]) == len(b[4]) and all(torch.allclose(u, v, rtol=1e-4, atol=1e-6) for u, v in zip(a[4], b[4]))


def test_generated_branch_with_grad_is_never_batched():
    """generator step: the generated signal carries a gradient -> two calls (no data gradient is paid for the real half)"""
    torch.manual_seed(1)
    d = D.DiscriminatorP(2)
    calls = []
    fwd = lambda x: (calls.append(x.shape[0]), TR.disc_p_forward(d, x))[1]
    y, y_hat = torch.randn(2, 1, 600), torch.randn(2, 1, 600, requires_grad=True)
    TR.disc_pair(d, y, y_hat, True, fwd=fwd)
    assert calls == [2, 2]
    calls.clear()
    TR.disc_pair(d, y, y_hat.detach(), True, fwd=fwd)
    assert calls == [4]


def test_drop_in_classes_keep_the_public_layout_and_have_no_cpu_path():
    """state_dict keys / shapes of hifigan.models' discriminators (what a reference checkpoint's `_mpd.` / `_msd.` entries hold, cubegan.py:46-47);
    the forwards and the loss functions are the HIP ones: CPU tensors raise instead of silently running torch ops"""
    from ttscube_amd._lib import TTSCError
    mpd, msd = D.MultiPeriodDiscriminator(), D.MultiScaleDiscriminator()
    sd = mpd.state_dict()
    assert [d.period for d in mpd.discriminators] == [2, 3, 5, 7, 11]
    assert tuple(sd['discriminators.0.convs.0.weight_v'].shape) == (32, 1, 5, 1) and tuple(sd['discriminators.4.conv_post.weight_g'].shape) == (1, 1, 1, 1)
    assert tuple(sd['discriminators.2.convs.4.weight_v'].shape) == (1024, 1024, 5, 1)
    sd = msd.state_dict()
    assert {'discriminators.0.convs.0.weight_orig', 'discriminators.0.convs.0.weight_u', 'discriminators.0.convs.0.weight_v',
            'discriminators.1.convs.1.weight_g', 'discriminators.2.conv_post.bias'} <= set(sd)
    assert tuple(sd['discriminators.1.convs.3.weight_v'].shape) == (512, 16, 41)
    y = torch.zeros(1, 1, 600)
    for mod in (mpd, msd, mpd.discriminators[0], msd.discriminators[1]):
        with pytest.raises(TTSCError, match='no CPU path'):
            mod(y, y) if mod in (mpd, msd) else mod(y)
    assert D.feature_loss.__module__.endswith('losses_hip') and D.mel_spectrogram.__module__.endswith('io_utils.melspec')
