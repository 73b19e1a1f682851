"""Import helpers for the READ-ONLY reference at /root/reference (build container only; never on the GPU box).
Registers the 10-line `pytorch_lightning` stub SURVEY.md §8c verified sufficient, then the reference modules
import unmodified."""
import sys
import types

REF = '/root/reference'


def setup():
    if 'pytorch_lightning' not in sys.modules:
        import torch
        pl = types.ModuleType('pytorch_lightning')

        class LightningModule(torch.nn.Module):
            def log(self, *a, **k):
                pass

            def log_dict(self, *a, **k):
                pass

        pl.LightningModule = LightningModule
        cb = types.ModuleType('pytorch_lightning.callbacks')
        cb.Callback = object
        pl.callbacks = cb
        pl.Trainer = object
        sys.modules['pytorch_lightning'] = pl
        sys.modules['pytorch_lightning.callbacks'] = cb
    if REF not in sys.path:
        sys.path.insert(0, REF)
