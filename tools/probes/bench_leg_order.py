"""Why does bench.py's cubegan_training_step_b16 leg read ~7 ms more than tools/bench_cubegan_step.py --ragged on the same box?
Runs the leg's loop after a chosen list of things bench.py does in front of it.
  python tools/probes/bench_leg_order.py [headline] [single] [selfcheck] [probe] [wavernn] [threads8] [gcfreeze] [gcoff] [emptycache]"""
import gc
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from oracle import hifigan_ref as R
    from ttscube_amd.hifigan.env import AttrDict
    from ttscube_amd.hifigan.models import Generator
    pre = sys.argv[1:]
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    keep = []
    if 'headline' in pre or 'selfcheck' in pre:
        h = dict(R.CONFIG_V1)
        sd = R.synthetic_state_dict(h, seed=1234)
        g = Generator(AttrDict(h))
        g.load_state_dict(sd)
        g = g.to(dev).eval()
        mel = R.synthetic_mel(64 if 'headline' in pre else 1, 800 if 'headline' in pre else 60, seed=1234).to(dev)
        with torch.no_grad():
            for _ in range(9 if 'headline' in pre else 1):
                out = g(mel)
        torch.cuda.synchronize()
        if 'selfcheck' in pre:
            bench.self_check(g, mel, out, h, sd)
        keep += [g, mel, out]
    if 'single' in pre:
        h = dict(R.CONFIG_V1)
        sd = R.synthetic_state_dict(h, seed=1234)
        g1 = Generator(AttrDict(h))
        g1.load_state_dict(sd)
        g1 = g1.to(dev).eval()
        mel1 = R.synthetic_mel(1, 300, seed=77).to(dev)
        bench.time_forward(g1, mel1, 50, 20, check='deferred')
        bench.time_forward(g1, mel1, 50, 5)
        keep += [g1, mel1]
    if 'probe' in pre:
        import ctypes as C
        from ttscube_amd import _lib
        for mode in (0, 1, 2):
            tf = C.c_double(0.0)
            _lib.check(_lib.lib().ttsc_probe_mfma_tflops(mode, 60.0, C.byref(tf), _lib.current_stream()), 'probe')
    if 'wavernn' in pre:
        from oracle import wavernn_ref as WO
        from ttscube_amd.networks.modules import WaveRNN
        wsd = WO.synthetic_state_dict(H=512, num_layers=1, use_lowres=True, seed=5)
        net = WaveRNN(num_layers=1, layer_size=512, upsample=240, upsample_low=10, use_lowres=True, output='mulaw')
        net.load_state_dict({k: torch.from_numpy(v) for k, v in wsd.items()}, strict=True)
        net = net.to(dev).eval()
        wm, wx = WO.synthetic_inputs(256, 100, seed=6)
        net.decode({'mel': torch.from_numpy(wm), 'x_low': torch.from_numpy(wx)}, mode='philox', seed=2)
        torch.cuda.synchronize()
        keep.append(net)
    if 'threads8' in pre:
        torch.set_num_threads(8)
    if 'emptycache' in pre:
        torch.cuda.empty_cache()
    from ttscube_amd.io_utils.io_cubegan import CubeganCollate
    from ttscube_amd.io_utils.synthetic import synthetic_encodings, synthetic_examples
    from ttscube_amd.networks.cubegan import Cubegan
    enc = synthetic_encodings()
    torch.manual_seed(1234)
    model = Cubegan(enc, conditioning=None, train=True).to(dev)
    model.train()
    batch = CubeganCollate(enc).collate_fn(list(synthetic_examples(16, 777, min_ph=30, max_ph=50)))
    crop = random.Random(99)
    for _ in range(4):
        out = model.training_step(batch, 0, rng=crop)
    torch.cuda.synchronize()
    if 'gcfreeze' in pre:
        gc.collect()
        gc.freeze()
    if 'gcoff' in pre:
        gc.disable()
    c0 = [s['collections'] for s in gc.get_stats()]
    t0 = time.perf_counter()
    for _ in range(8):
        out = model.training_step(batch, 0, rng=crop)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    c1 = [s['collections'] for s in gc.get_stats()]
    print('%-40s %.1f ms/step   gc collections in the timed steps %s  threads %d  objects %d' %
          (' '.join(pre) or '(nothing first)', dt * 1e3, [b - a for a, b in zip(c0, c1)], torch.get_num_threads(), len(gc.get_objects())))


if __name__ == '__main__':
    main()
