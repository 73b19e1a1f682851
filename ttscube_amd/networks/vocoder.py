"""Mirror of cube/networks/vocoder.py: ``CubenetVocoder`` = low-res WaveRNN (2.4 kHz) + high-res WaveRNN (24 kHz)
conditioned on the low-res signal, with the reference's chunk-folded inference (vocoder.py:96-131) and checkpoint
layout (`_wavernn_hr.` / `_wavernn_lr.` prefixes; train_vocoder.py:36-59)."""
import numpy as np
import torch
import torch.nn as nn

from .modules import WaveRNN


class CubenetVocoder(nn.Module):
    def __init__(self, num_layers_lr: int = 2, layer_size_lr: int = 512, num_layers_hr: int = 2, layer_size_hr: int = 512,
                 upsample=100, upsample_low=10, learning_rate=1e-4, output='mol'):
        super().__init__()
        self._learning_rate = learning_rate
        self._wavernn_hr = WaveRNN(num_layers=num_layers_hr, layer_size=layer_size_hr, upsample=upsample, use_lowres=True,
                                   upsample_low=upsample_low, learning_rate=learning_rate, output=output)
        self._wavernn_lr = WaveRNN(num_layers=num_layers_lr, layer_size=layer_size_lr, upsample=upsample // upsample_low,
                                   use_lowres=False, learning_rate=learning_rate, output=output)
        self._val_loss_hr = 9999
        self._val_loss_lr = 9999
        self.automatic_optimization = False
        self._global_step = 0
        self._upsample = upsample
        self._upsample_low = upsample_low

    def forward(self, X):
        if 'x' in X:
            return self._train(X)
        return self._inference(X)

    def _train(self, X):
        from .training import wavernn_loss
        loss_hr = wavernn_loss(self._wavernn_hr, {'x': X['x'], 'x_low': X['x_low'], 'mel': X['mel']})
        loss_lr = wavernn_loss(self._wavernn_lr, {'x': X['x_low'], 'mel': X['mel']})
        return {'lr': loss_lr, 'hr': loss_hr, 'loss': (loss_hr + loss_lr) / 2}

    def _inference(self, X, num_batches=20, **kw):
        """vocoder.py:96-107: lr net over the utterance, fold the hr problem into `num_batches` chunks (time -> batch),
        hr net over the folded batch, drop each chunk's warm-up prefix and concatenate.  Returns (x_lr, x_hr) numpy."""
        with torch.no_grad():
            dev = self._wavernn_lr._get_device()
            mel = X['mel'].to(dev).float()
            _, x_lr, _ = self._wavernn_lr.decode({'mel': mel}, **kw)          # [B, 24T] on device
            folded = self._inference_batch(mel, x_lr, num_batches=num_batches)
            _, batched_x_hr, _ = self._wavernn_hr.decode(folded, **kw)        # [nb, (T/nb+1)*240 | ...]
            x_hr = self._compose_batched_inference(batched_x_hr)
        return x_lr.unsqueeze(2).cpu().numpy(), x_hr.cpu().numpy()

    def _compose_batched_inference(self, batched_x):
        batched_x = batched_x[:, self._upsample:]
        return batched_x.reshape(1, -1)

    def _inference_batch(self, mel, x_low, num_batches=5):
        """vocoder.py:113-131 on the device (torch indexing = data movement only): each chunk gets a 1-frame mel
        prefix (pad value -5 for chunk 0) and a `upsample_low`-sample x_low prefix (zeros for chunk 0)."""
        if mel.shape[1] < num_batches:
            num_batches = mel.shape[1]
        mel = mel[:, :mel.shape[1] // num_batches * num_batches]
        x_low = x_low[:, :x_low.shape[1] // num_batches * num_batches]
        mel_split = mel.reshape(num_batches, -1, mel.shape[2])
        x_low_split = x_low.reshape(num_batches, -1)
        m = torch.full((mel_split.shape[0], mel_split.shape[1] + 1, mel_split.shape[2]), -5.0, dtype=torch.float32,
                       device=mel.device)
        m[:, 1:, :] = mel_split
        m[1:, 0, :] = mel_split[:-1, -1, :]
        xl = torch.zeros((x_low_split.shape[0], x_low_split.shape[1] + self._upsample_low), dtype=torch.float32,
                         device=mel.device)
        xl[:, self._upsample_low:] = x_low_split
        xl[1:, 0:self._upsample_low] = x_low_split[:-1, -self._upsample_low:]
        return {'mel': m, 'x_low': xl}

    # ---- the LightningModule surface of vocoder.py:133-176 (pl.Trainer.fit(model) in the reference's scripts/train_vocoder.py) ----
    def configure_optimizers(self):
        """vocoder.py:169-173: two Adam optimizers (low-resolution net first)"""
        self._optimizers = [torch.optim.Adam(self._wavernn_lr.parameters(), lr=self._learning_rate),
                            torch.optim.Adam(self._wavernn_hr.parameters(), lr=self._learning_rate)]
        return self._optimizers

    def optimizers(self):
        if getattr(self, '_optimizers', None) is None:
            self.configure_optimizers()
        return self._optimizers

    def set_gradient_exchange(self, reducers):
        self._reducers = reducers

    def log_dict(self, d, **kw):
        fn = getattr(self, '_log_fn', None)
        if fn is not None:
            fn(d)

    def log(self, name, value, **kw):
        self.log_dict({name: value})

    def training_step(self, batch, batch_idx=None):
        """vocoder.py:136-156: both networks' teacher-forced CE losses, clip_grad_norm 5, Adam x 2, lr decay — on the HIP GRU / GEMM / convolution
        kernels behind autograd (networks/training.py::vocoder_training_step)"""
        from .training import vocoder_training_step
        out = vocoder_training_step(self, batch, self.optimizers(), getattr(self, '_reducers', None))
        self.log_dict(out, prog_bar=True)
        return out

    def validation_step(self, batch, batch_idx=None):
        """vocoder.py:133-134"""
        with torch.no_grad():
            return self.forward(batch)

    def validation_epoch_end(self, outputs) -> None:
        """vocoder.py:158-165"""
        loss_lr = sum(float(x['lr']) for x in outputs) / len(outputs)
        loss_hr = sum(float(x['hr']) for x in outputs) / len(outputs)
        self.log('val_loss', (loss_hr + loss_lr) / 2)
        self._val_loss_hr = loss_hr
        self._val_loss_lr = loss_lr

    @torch.jit.ignore
    def save(self, path):
        torch.save(self.state_dict(), path)

    @torch.jit.ignore
    def load(self, path):
        self.load_state_dict(torch.load(path, map_location='cpu'))

    def _compute_lr(self, initial_lr, delta, step):
        return initial_lr / (1 + delta * step)
