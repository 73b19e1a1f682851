"""The all-torch formulation of the training steps — TEST INFRASTRUCTURE, never imported by the product package.

`install(monkeypatch)` swaps every native building block of ttscube_amd.networks.training (HIP autograd functions, GAN-loss kernels,
flat-arena AdamW, native discriminators, MFMA Linears) for its torch-op counterpart on the SAME parameter tensors, so that a test can
run the product's step function twice — native and all-torch — and compare losses / gradients
(reference semantics: cube/networks/cubegan.py:85-189, cube/networks/modules.py:505-563)."""
import torch
import torch.nn.functional as F


def _wn(l):
    """live weight-norm: w = g * v / ||v|| (so that gradients reach weight_g and weight_v)."""
    if hasattr(l, 'weight'):
        return l.weight
    v, g = l.weight_v, l.weight_g
    return g * v / v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, 1, 1)


def generator_forward_train(gen, x):
    """HiFi-GAN generator forward as torch ops: the autograd reference of hifigan/autograd.py::generator_forward_with_grad."""
    from ttscube_amd.hifigan.models import ResBlock1
    h = gen.h
    x = F.conv1d(x, _wn(gen.conv_pre), gen.conv_pre.bias, padding=3)
    nk = gen.num_kernels
    for i, (u, k) in enumerate(zip(h['upsample_rates'], h['upsample_kernel_sizes'])):
        x = F.leaky_relu(x, 0.1)
        x = F.conv_transpose1d(x, _wn(gen.ups[i]), gen.ups[i].bias, stride=u, padding=(k - u) // 2)
        xs = None
        for j in range(nk):
            rb = gen.resblocks[i * nk + j]
            kr, ds = h['resblock_kernel_sizes'][j], h['resblock_dilation_sizes'][j]
            r = x
            if isinstance(rb, ResBlock1):
                for c1, c2, d in zip(rb.convs1, rb.convs2, ds):
                    xt = F.conv1d(F.leaky_relu(r, 0.1), _wn(c1), c1.bias, dilation=d, padding=d * (kr - 1) // 2)
                    xt = F.conv1d(F.leaky_relu(xt, 0.1), _wn(c2), c2.bias, padding=(kr - 1) // 2)
                    r = xt + r
            else:
                for c, d in zip(rb.convs, ds):
                    r = F.conv1d(F.leaky_relu(r, 0.1), _wn(c), c.bias, dilation=d, padding=d * (kr - 1) // 2) + r
            xs = r if xs is None else xs + r
        x = xs / nk
    x = F.leaky_relu(x)
    x = F.conv1d(x, _wn(gen.conv_post), gen.conv_post.bias, padding=3)
    return torch.tanh(x)


def _text_ops(lang):
    def cnn(name, h):
        for layer in getattr(lang, name):
            if hasattr(layer, 'conv'):
                h = torch.tanh(F.conv1d(h, layer.conv.weight, layer.conv.bias, padding=1))
        return h
    return (lambda emb, idx: emb(idx)), F.linear, cnn


def _make_adamw(params, lr):
    return torch.optim.AdamW(params, lr, betas=[0.8, 0.99], fused=all(p.is_cuda for p in params))


def _gan_loss_fns():
    from ttscube_amd.hifigan.discriminators import discriminator_loss, feature_loss, generator_loss
    return discriminator_loss, feature_loss, generator_loss


def _discriminator_fns(model):
    return (lambda a_, b_, fm=True: model._mpd(a_, b_)), (lambda a_, b_, fm=True: model._msd(a_, b_))


def _lowres_features(net, hidden):
    for conv in net._lowres_conv:
        hidden = torch.tanh(F.conv1d(hidden, conv.conv.weight, conv.conv.bias, padding=3))
    return hidden


def _output_linears(net, hidden):
    pre = torch.tanh(F.linear(hidden, net._preoutput.linear_layer.weight, net._preoutput.linear_layer.bias))
    return F.linear(pre, net._output.linear_layer.weight, net._output.linear_layer.bias)


def install(monkeypatch, recurrences=True, mel=True):
    """every native piece of ttscube_amd.networks.training -> torch ops (undone by monkeypatch at the end of the test)"""
    from ttscube_amd.hifigan import discriminators as D
    from ttscube_amd.io_utils import melspec as MS
    from ttscube_amd.networks import training as T
    for name in ('_text_ops', '_make_adamw', '_gan_loss_fns', '_discriminator_fns', '_lowres_features', '_output_linears'):
        monkeypatch.setattr(T, name, globals()[name])
    monkeypatch.setattr(T, 'generator_forward_with_grad', generator_forward_train)
    if recurrences:
        monkeypatch.setattr(T, 'lstm_forward_train', lambda rnn, x: rnn(x)[0])
        monkeypatch.setattr(T, 'gru_forward_train', lambda rnn, x: rnn(x)[0])
    if mel:
        monkeypatch.setattr(MS, 'mel_spectrogram', D.mel_spectrogram)
