"""hifigan/discriminators.py: a weight-normed sub-discriminator may see real and generated audio as one batch (the discriminator
step of `cubegan_training_step`) — outputs, feature maps, loss and parameter gradients must equal two separate calls."""
import torch

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
            r, fr, g, fg = D._pair(d, y, y_hat, batched)
            loss, _, _ = D.discriminator_loss([r], [g])
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
    orig = d.forward
    d.forward = lambda x: (calls.append(x.shape[0]), orig(x))[1]
    y, y_hat = torch.randn(2, 1, 600), torch.randn(2, 1, 600, requires_grad=True)
    D._pair(d, y, y_hat, True)
    assert calls == [2, 2]
    calls.clear()
    D._pair(d, y, y_hat.detach(), True)
    assert calls == [4]


def test_losses_return_tensors_not_host_scalars():
    """no .item() in the GAN losses: 16 host syncs per training step would drain the launch queue"""
    a, b = [torch.randn(2, 10)], [torch.randn(2, 10)]
    _, r, g = D.discriminator_loss(a, b)
    assert all(torch.is_tensor(v) for v in r + g)
