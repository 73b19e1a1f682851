"""Golden vectors for the WaveRNN path, produced by the REFERENCE ITSELF (imported from /root/reference).

    python tools/gen_golden_wavernn.py   ->  tests/golden/mulaw_lut.npy, mulaw_kat.npz, wavernn_*.npz

The reference's sampler (`Categorical(logits).sample()`, cube/networks/loss.py:227-230) draws
`E = empty_like(probs).exponential_(1)` per step from torch's global CPU generator and returns argmax(probs/E).
We run `WaveRNN._inference` UNPATCHED after torch.manual_seed(s), replay the same exponential stream after the same
seed, and check the replay explains every emitted sample before saving g = -log(E) as the injected Gumbel noise.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import ref_import  # noqa: E402

ref_import.setup()
from cube.networks.loss import MOLOutput, MULAWOutput, RAWOutput  # noqa: E402
from cube.networks.modules import WaveRNN  # noqa: E402
from cube.networks.vocoder import CubenetVocoder  # noqa: E402
from oracle import wavernn_ref as O  # noqa: E402  (synthetic weights/inputs only)

OUT = os.path.join(ROOT, 'tests', 'golden')


def gen_mulaw():
    m = MULAWOutput()
    lut = m.decode(torch.arange(256)).numpy().astype(np.float32)
    np.save(os.path.join(OUT, 'mulaw_lut.npy'), lut)
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(4096, generator=g) * 2 - 1).float()
    x = torch.cat([x, torch.tensor([1.0, 0.9, 0.0, -0.9, -1.0, 1e-5, -1e-5, 0.5, -0.5])])
    np.savez(os.path.join(OUT, 'mulaw_kat.npz'), x=x.numpy(), enc=m.encode(x).numpy(), enc_raw=RAWOutput().encode(x).numpy(),
             dec_raw=RAWOutput().decode(torch.arange(256).float()).numpy())
    print('mulaw lut', lut[[0, 1, 127, 128, 254, 255]])


def run_case(name, H, N, use_lowres, B, T, seed, output='mulaw'):
    up = 240 if use_lowres else 24
    torch.manual_seed(0)
    net = WaveRNN(num_layers=N, layer_size=H, upsample=up, upsample_low=10, use_lowres=use_lowres, output=output)
    sd = O.synthetic_state_dict(H=H, num_layers=N, use_lowres=use_lowres, seed=seed)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net.eval()
    mel, x_low = O.synthetic_inputs(B, T, seed=seed + 1)
    X = {'mel': torch.from_numpy(mel)}
    if use_lowres:
        X['x_low'] = torch.from_numpy(x_low)
    torch.manual_seed(seed)
    wav = net._inference(dict(X))  # numpy [B, L, 1]
    wav = wav.reshape(B, -1)
    L = wav.shape[1]
    # replay the exponential stream the sampler consumed
    torch.manual_seed(seed)
    E = torch.stack([torch.empty(B, 256).exponential_(1) for _ in range(L)], dim=1)  # [B, L, 256]
    # verify the replay by re-running the reference with a sampler that consumes the replayed E
    fn = net._output_functions
    step = [0]
    orig = fn.sample

    def patched(y):
        probs = torch.softmax(y, dim=-1)
        q = probs / E[:, step[0]].unsqueeze(1)
        step[0] += 1
        return fn.decode(torch.argmax(q, dim=-1))

    fn.sample = patched
    wav2 = net._inference(dict(X)).reshape(B, -1)
    fn.sample = orig
    assert np.array_equal(wav, wav2), 'exponential-stream replay does not reproduce the reference samples'
    # teacher-forced logits: WaveRNN._train_forward on random audio (modules.py:505-539, shift as in 553-558)
    g = torch.Generator().manual_seed(seed + 2)
    audio = (torch.rand(B, L, generator=g) * 2 - 1).float()
    xin = torch.nn.functional.pad(audio[:, :-1], (1, 0), value=0)
    Xt = dict(X)
    Xt['x'] = xin
    with torch.no_grad():
        logits = net._train_forward(Xt).numpy()
        loss = float(fn.loss(torch.from_numpy(logits), audio))
    blob = dict(H=H, N=N, use_lowres=int(use_lowres), B=B, T=T, seed=seed, output=output, mel=mel, x_low=x_low,
                gumbel=(-torch.log(E)).numpy().astype(np.float32), wav=wav.astype(np.float32),
                idx=fn.encode(torch.from_numpy(wav)).numpy().astype(np.uint8), audio=audio.numpy(),
                logits_tf=logits.astype(np.float32), loss_tf=np.float32(loss))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **blob)
    print(name, 'L', L, 'wav rms', float(np.sqrt((wav ** 2).mean())), 'uniq idx', len(np.unique(blob['idx'])))


def run_case_continuous(name, H, N, use_lowres, B, T, seed, output):
    """MOL / Gaussian / Beta outputs (loss.py:35-215; 'mol' is the reference default, modules.py:398).  The reference draws
    uniforms / normals from torch's global CPU generator inside sample(); for 'mol' and 'gm' we replay that stream after the
    same seed, check that the replay reproduces every sample, and store the random TERMS the reference added:
      mol: 10 Gumbel terms -log(-log(u1)) + 1 logistic term log(u2) - log(1 - u2)   -> noise [B, L, 11]
      gm : z * 0.8                                                                   -> noise [B, L, 1]
    Beta sampling goes through torch's rejection sampler (not replayable): its golden holds the teacher-forced outputs and
    the loss only; the sampler is pinned distributionally in tests/test_oracle_wavernn.py."""
    up = 240 if use_lowres else 24
    S = O.SAMPLE_SIZE[output]
    torch.manual_seed(0)
    net = WaveRNN(num_layers=N, layer_size=H, upsample=up, upsample_low=10, use_lowres=use_lowres, output=output)
    sd = O.synthetic_state_dict(H=H, num_layers=N, use_lowres=use_lowres, seed=seed, S=S)
    if output == 'mol':     # keep the log-scales in a range where samples are not all clamped to +-1
        sd['_output.linear_layer.weight'][20:] *= 0.25
        sd['_output.linear_layer.bias'][20:] = sd['_output.linear_layer.bias'][20:] * 0.5 - 3.0
        sd['_output.linear_layer.weight'][10:20] *= 0.5
    if output == 'gm':
        sd['_output.linear_layer.weight'][1:] *= 0.25
        sd['_output.linear_layer.bias'][1:] -= 2.5
        sd['_output.linear_layer.weight'][:1] *= 0.3
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net.eval()
    mel, x_low = O.synthetic_inputs(B, T, seed=seed + 1)
    X = {'mel': torch.from_numpy(mel)}
    if use_lowres:
        X['x_low'] = torch.from_numpy(x_low)
    fn = net._output_functions
    blob = dict(H=H, N=N, use_lowres=int(use_lowres), B=B, T=T, seed=seed, output=output, mel=mel, x_low=x_low,
                out_w=sd['_output.linear_layer.weight'], out_b=sd['_output.linear_layer.bias'])
    L = T * up
    if output in ('mol', 'gm'):
        torch.manual_seed(seed)
        wav = net._inference(dict(X)).reshape(B, -1)
        L = wav.shape[1]
        torch.manual_seed(seed)
        if output == 'mol':
            u1, u2 = [], []
            for _ in range(L):
                u1.append(torch.empty(B, 1, 10).uniform_(1e-5, 1 - 1e-5))
                u2.append(torch.empty(B, 1).uniform_(1e-5, 1.0 - 1e-5))
            u1, u2 = torch.cat(u1, dim=1), torch.cat(u2, dim=1)              # [B, L, 10], [B, L]
            noise = torch.cat([-torch.log(-torch.log(u1)), (torch.log(u2) - torch.log(1. - u2)).unsqueeze(2)], dim=2)
        else:
            noise = torch.cat([torch.randn((B, 1, 1)) * 0.8 for _ in range(L)], dim=1)   # [B, L, 1]
        step = [0]
        orig = fn.sample
        kidx = np.zeros((B, L), dtype=np.uint8)

        def patched(y, *a, **k):
            t = step[0]
            step[0] += 1
            if output == 'mol':
                temp = y[:, :, :10].data + noise[:, t, :10].unsqueeze(1)
                _, am = temp.max(dim=-1)
                kidx[:, t] = am[:, 0].numpy()
                oh = torch.nn.functional.one_hot(am, 10).float()
                means = torch.sum(y[:, :, 10:20] * oh, dim=-1)
                ls = torch.clamp(torch.sum(y[:, :, 20:30] * oh, dim=-1), min=float(np.log(1e-14)))
                x = means + torch.exp(ls) * noise[:, t, 10].unsqueeze(1)
                return torch.clamp(torch.clamp(x, min=-1.), max=1.)
            return (y[:, :, 0].unsqueeze(2) + noise[:, t].unsqueeze(1) * torch.exp(y[:, :, 1].unsqueeze(2))).squeeze(1)

        fn.sample = patched
        wav2 = net._inference(dict(X)).reshape(B, -1)
        fn.sample = orig
        assert np.array_equal(wav, wav2), 'random-stream replay does not reproduce the reference samples'
        blob.update(noise=noise.numpy().astype(np.float32), wav=wav.astype(np.float32), idx=kidx)
    g = torch.Generator().manual_seed(seed + 2)
    audio = (torch.rand(B, L, generator=g) * 1.6 - 0.8).float()
    xin = torch.nn.functional.pad(audio[:, :-1], (1, 0), value=0)
    Xt = dict(X)
    Xt['x'] = xin
    with torch.no_grad():
        logits = net._train_forward(Xt)
        loss = float(fn.loss(logits, audio))
    blob.update(audio=audio.numpy(), logits_tf=logits.numpy().astype(np.float32), loss_tf=np.float32(loss))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **blob)
    print(name, 'L', L, 'loss', loss, ('wav rms %.3f clamp share %.3f' % (float(np.sqrt((blob['wav'] ** 2).mean())),
                                                                        float((np.abs(blob['wav']) >= 1).mean()))) if 'wav' in blob else '')


def gen_vocoder_fold():
    """CubenetVocoder._inference_batch / _compose_batched_inference shapes + values on T=40 (vocoder.py:109-131)."""
    torch.manual_seed(0)
    voc = CubenetVocoder(num_layers_lr=1, layer_size_lr=16, num_layers_hr=1, layer_size_hr=16, upsample=240,
                         upsample_low=10, output='mulaw')
    mel, _ = O.synthetic_inputs(1, 43, seed=77)
    x_low = np.random.RandomState(3).uniform(-1, 1, size=(1, 43 * 24)).astype(np.float32)
    fold = voc._inference_batch(torch.from_numpy(mel), torch.from_numpy(x_low), num_batches=20)
    hr = np.random.RandomState(4).uniform(-1, 1, size=(20, fold['x_low'].shape[1] * 10)).astype(np.float32)
    comp = voc._compose_batched_inference(hr)
    np.savez_compressed(os.path.join(OUT, 'vocoder_fold.npz'), mel=mel, x_low=x_low, fold_mel=fold['mel'].numpy(),
                        fold_x_low=fold['x_low'].numpy(), hr=hr, composed=comp)
    print('fold', tuple(fold['mel'].shape), tuple(fold['x_low'].shape), comp.shape)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    gen_mulaw()
    gen_vocoder_fold()
    run_case('wavernn_hr_h64_n1', 64, 1, True, 2, 2, 11)
    run_case('wavernn_hr_h64_n2', 64, 2, True, 2, 1, 12)
    run_case('wavernn_lr_h64_n1', 64, 1, False, 2, 6, 13)
    run_case('wavernn_hr_h512_n1', 512, 1, True, 1, 1, 14)
    run_case('wavernn_hr_h64_raw', 64, 1, True, 1, 1, 15, output='raw')
    run_case_continuous('wavernn_hr_h64_mol', 64, 1, True, 2, 2, 16, 'mol')
    run_case_continuous('wavernn_hr_h512_mol', 512, 2, True, 1, 1, 17, 'mol')   # the reference's default constructor arguments
    run_case_continuous('wavernn_lr_h64_gm', 64, 1, False, 2, 5, 18, 'gm')
    run_case_continuous('wavernn_hr_h64_beta', 64, 1, True, 2, 1, 19, 'beta')
