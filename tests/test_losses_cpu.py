"""CPU: ttscube_amd/networks/loss.py (own formulation) against known answers captured from the reference's
cube/networks/loss.py (tools/gen_golden_losses.py -> tests/golden/losses_kat.npz)."""
import os

import numpy as np
import torch

from ttscube_amd.networks import loss as PL


def _rel(a, b):
    return abs(float(a) - float(b)) / max(1e-12, abs(float(b)))


def test_losses_and_codecs_match_reference_known_answers(golden_dir):
    z = np.load(os.path.join(golden_dir, 'losses_kat.npz'))
    y = torch.from_numpy(z['y'])
    assert _rel(PL.MOLOutput().loss(torch.from_numpy(z['mol_in']), y), z['mol_loss']) < 2e-6
    assert _rel(PL.GaussianOutput().loss(torch.from_numpy(z['gm_in']), y), z['gm_loss']) < 2e-6
    assert _rel(PL.BetaOutput().loss(torch.from_numpy(z['beta_in']), y), z['beta_loss']) < 1e-5
    lg = torch.from_numpy(z['cls_in'])
    m, r = PL.MULAWOutput(), PL.RAWOutput()
    assert _rel(m.loss(lg, y), z['mulaw_loss']) < 2e-6 and _rel(r.loss(lg, y), z['raw_loss']) < 2e-6
    assert np.array_equal(m.encode(y).numpy(), z['mulaw_enc']) and np.array_equal(m.encode(z['y']), z['mulaw_enc_np'])
    codes = np.arange(256)
    assert np.array_equal(m.decode(torch.from_numpy(codes)).numpy(), z['mulaw_dec'])
    assert np.array_equal(m.decode(codes.astype(np.float64)), z['mulaw_dec_np'])
    assert np.array_equal(r.encode(y).numpy(), z['raw_enc']) and np.array_equal(r.decode(torch.from_numpy(codes).float()).numpy(), z['raw_dec'])
    for name, cls in (('mol', PL.MOLOutput), ('gm', PL.GaussianOutput), ('beta', PL.BetaOutput), ('mulaw', PL.MULAWOutput), ('raw', PL.RAWOutput)):
        o = cls()
        assert [o.sample_size, *o.stats] == list(z[name + '_meta'])


def test_mol_loss_gradients_are_finite_on_every_branch(golden_dir):
    z = np.load(os.path.join(golden_dir, 'losses_kat.npz'))
    x = torch.from_numpy(z['mol_in']).requires_grad_(True)
    PL.MOLOutput().loss(x, torch.from_numpy(z['y'])).backward()
    assert bool(torch.isfinite(x.grad).all()) and float(x.grad.abs().sum()) > 0
