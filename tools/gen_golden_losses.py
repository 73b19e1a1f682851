"""Build container only: imports the reference's output-distribution classes (cube/networks/loss.py) and stores known answers for
their `loss` / `encode` / `decode` on seeded inputs -> tests/golden/losses_kat.npz.  The fixture is data (inputs + the reference's
outputs); tests/test_losses_cpu.py holds ttscube_amd/networks/loss.py to it."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import  # noqa: E402

ref_import.setup()
from cube.networks import loss as RL  # noqa: E402

rs = np.random.RandomState(20260927)
B, L = 3, 96
out = {}
y = np.clip(rs.randn(B, L) * 0.4, -1, 1).astype(np.float32)
y[0, :4] = [-1.0, 1.0, -0.9995, 0.9995]     # both edge branches of the MOL likelihood
out['y'] = y
yt = torch.from_numpy(y)
# MOL: 10 mixtures; a few very narrow components so that the bin mass underflows (the density branch)
mol = (rs.randn(B, L, 30) * 0.7).astype(np.float32)
mol[:, ::5, 20:] -= 9.0
out['mol_in'] = mol
out['mol_loss'] = np.float64(RL.MOLOutput().loss(torch.from_numpy(mol), yt).item())
gm = (rs.randn(B, L, 2) * 0.5).astype(np.float32)
gm[0, 0] = [y[0, 0] + 1e-6, -20.0]   # below the log-std floor
out['gm_in'] = gm
out['gm_loss'] = np.float64(RL.GaussianOutput().loss(torch.from_numpy(gm), yt).item())
bt = (rs.randn(B, L, 2) * 0.5 + 0.5).astype(np.float32)
out['beta_in'] = bt
out['beta_loss'] = np.float64(RL.BetaOutput().loss(torch.from_numpy(bt), yt).item())
lg = rs.randn(B, L, 256).astype(np.float32)
out['cls_in'] = lg
m = RL.MULAWOutput()
out['mulaw_loss'] = np.float64(m.loss(torch.from_numpy(lg), yt).item())
out['mulaw_enc'] = m.encode(yt).numpy()
out['mulaw_enc_np'] = m.encode(y)
codes = np.arange(256)
out['mulaw_dec'] = m.decode(torch.from_numpy(codes)).numpy()
out['mulaw_dec_np'] = m.decode(codes.astype(np.float64))
r = RL.RAWOutput()
out['raw_loss'] = np.float64(r.loss(torch.from_numpy(lg), yt).item())
out['raw_enc'] = r.encode(yt).numpy()
out['raw_dec'] = r.decode(torch.from_numpy(codes).float()).numpy()
for name, cls in (('mol', RL.MOLOutput), ('gm', RL.GaussianOutput), ('beta', RL.BetaOutput), ('mulaw', RL.MULAWOutput), ('raw', RL.RAWOutput)):
    o = cls()
    out[name + '_meta'] = np.array([o.sample_size, *o.stats], dtype=np.float64)
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'losses_kat.npz')
np.savez_compressed(dst, **out)
print('wrote', dst, {k: (v.shape if hasattr(v, 'shape') and v.shape else float(v)) for k, v in out.items() if 'loss' in k})
