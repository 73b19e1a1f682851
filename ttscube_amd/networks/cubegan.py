"""Mirror of cube/networks/cubegan.py: ``Cubegan`` = Languasito2 (text -> 80-d conditioning) + HiFi-GAN Generator.
Same constructor, ``inference`` / ``forward`` / ``load`` (strict=False) / ``save`` / ``get_device`` and state_dict
prefixes (`_generator.`, `_languasito.`, `_mpd.`, `_msd.`, `_dummy.`).  The generator config is resolved from the JSON
shipped in this package (same keys as the reference's CWD-relative `hifigan/config_v1.json`, cubegan.py:41), or from a
`hifigan/config_v1.json` in the CWD when one exists."""
import json
import os

import torch

from .. import _lib
import torch.nn as nn

from ..hifigan.env import AttrDict
from ..hifigan.models import Generator
from .modules import Languasito2

_PIPE_TXT_PRIORITY = int(os.environ.get('TTSC_PIPE_TXT_PRIORITY', '-1'))   # inference_pipelined's text stream: -1 = high priority (own hardware-queue pool), 0 = reserved pool
_AUX = True if os.environ.get('TTSC_DUR_DEVICE', '1') == '0' else 'device'   # (measurement switch: 0 = durations read back inside Languasito2.inference)


def _load_hifigan_config():
    for p in ('hifigan/config_v1.json', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'hifigan', 'config_v1.json')):
        if os.path.exists(p):
            return AttrDict(json.load(open(p)))
    raise FileNotFoundError('hifigan/config_v1.json')


class Cubegan(nn.Module):
    def __init__(self, encodings, lr: float = 2e-4, conditioning=None, train=True):
        super().__init__()
        self._current_lr = lr
        self._learning_rate = lr
        self._global_step = 0
        self._encodings = encodings
        self._val_loss = 9999
        self._conditioning = conditioning
        self._loaded_optimizer_states = None
        cond_type = conditioning.split(':')[0] if conditioning not in (None, 'none') else None
        self._cond_type = cond_type
        self._generator = Generator(_load_hifigan_config())
        if train:
            from ..hifigan.discriminators import MultiPeriodDiscriminator, MultiScaleDiscriminator
            self._mpd = MultiPeriodDiscriminator()
            self._msd = MultiScaleDiscriminator()
        self._languasito = Languasito2(len(encodings.phon2int), len(encodings.speaker2int), encodings.max_pitch,
                                       encodings.max_duration, cond_type=cond_type)
        self._hf = None
        if cond_type == 'hf':
            raise NotImplementedError("conditioning='hf:<model>' needs a downloaded HuggingFace encoder (no network here); "
                                      "use conditioning=None, or 'fasttext'-style pre-computed X['x_words']")
        if train:
            self._dummy = nn.Linear(1, 1)
        self._loss_l1 = nn.L1Loss()
        self._loss_cross = nn.CrossEntropyLoss(ignore_index=int(max(encodings.max_pitch, encodings.max_duration) + 1))
        self.automatic_optimization = False

    def inference(self, X, return_lengths=False, timers=None, check='sync'):
        """cubegan.py:74-83: text -> conditioning (predicted durations/pitch) -> waveform [B,1,L] in (-1,1).
        With a padded batch (B>1, new capability) `return_lengths=True` also returns each utterance's sample count."""
        with torch.no_grad():
            cond, _, flens = self._languasito.inference(X, return_aux='device', check_status=False, timers=timers)
            if cond.shape[1] == 0:
                cond = torch.zeros((cond.shape[0], 1, cond.shape[2]), device=self.get_device())
                flens = [1] * cond.shape[0]
            # check: the generator's split-precision range guard ('sync' waits for this batch and reruns it if needed; 'deferred' lets a
            # pipelined caller keep the host ahead of the GPU — see hifigan.models.Generator.forward)
            wav = self._generator(cond.permute(0, 2, 1).contiguous(), frames=flens if cond.shape[0] > 1 else None, check=check)
            if timers is not None:   # phase boundaries for bench.py --mode e2e (see Languasito2.inference)
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                timers.append(('generator', ev))
        _lib.check_split_status_once('Cubegan.inference')   # the BiLSTM recurrences may run split over several workgroups
        if return_lengths:
            return wav, [self._generator.out_len(f) if f > 0 else 0 for f in flens]
        return wav

    def inference_pipelined(self, batches, check='deferred', lstm_group=4):
        """`inference` over a SEQUENCE of padded batches as a two-stage pipeline on two streams: the text / frame stacks of batch k + 1
        (latency-bound BiLSTM recurrences that occupy a few dozen CUs) run while the generator of batch k (which fills the chip) is still
        running.  Yields (wav [B,1,L], sample counts) per batch, in order; a yielded waveform is complete (its stream has been waited for).
        Same arithmetic as `inference` — per-batch results are bit-identical — only the overlap differs.  The reference is a B = 1 loop
        (cube/api.py:45-66); batching and pipelining are this implementation's."""
        dev = self.get_device()
        # the recurrences' stream gets the higher priority: their workgroups are few and must all be resident to make progress, the generator's are
        # thousands and short — the dispatcher should hand a freed CU to the recurrence first
        if _PIPE_TXT_PRIORITY == 0:
            # both pipeline streams out of the package's reserved pool (hifigan/streams.py::_reserve): no stream of another priority, i.e. no second
            # hardware-queue pool in the process
            from ..hifigan.streams import _side_streams, text_stream
            s_txt, s_gen = text_stream(dev), _side_streams(dev, 3)[2]     # (side streams 0 / 1 are the generator's branch streams)
        else:
            s_txt, s_gen = torch.cuda.Stream(device=dev, priority=_PIPE_TXT_PRIORITY), torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        s_txt.wait_stream(main)
        s_gen.wait_stream(main)
        pending = None

        def finish(p):
            wav, lens, done = p
            done.synchronize()
            return wav, lens

        try:                                          # (no_grad only around the compute: a `with` spanning a yield would leak into the consumer)
            for X in batches:
                # the caller may have produced X on ITS stream (host-to-device copies of the next batch while this generator was suspended)
                s_txt.wait_stream(main)
                for v in X.values():
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(s_txt)
                # recurrences packed `lstm_group` utterances per member group: they hold that many times fewer CUs (which the generator of the
                # previous batch is using) for a slightly longer step — the step time is hidden here, the CUs are not
                with torch.cuda.stream(s_txt), torch.no_grad(), _lib.lstm_group_size(lstm_group):
                    cond, _, flens = self._languasito.inference(X, return_aux=_AUX, check_status=False)   # (waits for ITS stream only: frame counts)
                    if cond.shape[1] == 0:
                        cond = torch.zeros((cond.shape[0], 1, cond.shape[2]), device=dev)
                        flens = [1] * cond.shape[0]
                    cond_t = cond.permute(0, 2, 1).contiguous()
                    ready = torch.cuda.Event()
                    ready.record(s_txt)
                cond_t.record_stream(s_gen)
                if pending is not None:               # hand the previous batch out while this one's generator is about to be queued
                    out = finish(pending)
                    pending = None
                    yield out
                with torch.cuda.stream(s_gen), torch.no_grad():
                    s_gen.wait_event(ready)
                    wav = self._generator(cond_t, frames=flens if cond_t.shape[0] > 1 else None, check=check)
                    done = torch.cuda.Event()
                    done.record(s_gen)
                wav.record_stream(main)
                pending = (wav, [self._generator.out_len(f) if f > 0 else 0 for f in flens], done)
            # the verdicts of the deferred range guard and of the split recurrences are collected BEFORE the last batch is handed out: a consumer
            # that stops pulling after the last result (zip, islice) never resumes this generator, so nothing may be left to do behind that yield
            last = finish(pending) if pending is not None else None
            pending = None
            with torch.cuda.stream(s_gen):
                self._generator.finish_range_check()
            _lib.check_split_status('Cubegan.inference_pipelined')
            if last is not None:
                yield last
        finally:                                      # also when the consumer stops early: whatever is still queued is ordered before the caller's stream
            main.wait_stream(s_gen)
            main.wait_stream(s_txt)
            if self._generator.range_check_pending():  # consumer stopped early: do not leave a stale verdict for the next, unrelated forward
                with torch.cuda.stream(s_gen):
                    tripped = self._generator.finish_range_check(raise_on_trip=False)
                if tripped:
                    # batches ALREADY handed out may hold non-finite audio; raising from a generator's clean-up (GeneratorExit) is not allowed,
                    # so the verdict is reported as a warning rather than swallowed
                    import warnings
                    warnings.warn('Cubegan.inference_pipelined was stopped early and its deferred range guard had tripped: audio already '
                                  'yielded by this call may be non-finite; rerun those batches with check="sync"', RuntimeWarning)

    def forward(self, X):
        """cubegan.py:65-72: forced alignment path (X carries y_frame2phone / y_pitch)."""
        from .training import languasito_forward
        with torch.no_grad():
            _, _, _, cond = languasito_forward(self._languasito, X)
            return self._generator(cond.permute(0, 2, 1).contiguous())

    # ---- the LightningModule surface the reference's trainer drives (scripts/train_cubegan.py:138-145 of the reference: pl.Trainer.fit(model)) ----
    # The step logic lives in networks/training.py (HIP kernels end to end); these methods are the reference's entry points onto it, usable with or
    # without a trainer object: `optimizers()` hands out the tuple `configure_optimizers()` built (what Lightning's own accessor returns under
    # manual optimisation), `log_dict` is a no-op unless a logger was attached with `set_logger`.
    def configure_optimizers(self):
        """cubegan.py:275-311: AdamW(0.8, 0.99) over the generator side, the discriminators and the text side + Adam(1e-6) on the dummy — as flat-arena
        HIP optimizers (optim.FlatAdamW, torch.optim.AdamW's state_dict layout); restores `.opt.last` states queued in `_loaded_optimizer_states`"""
        from .training import cubegan_configure_optimizers
        self._optimizers = cubegan_configure_optimizers(self)
        return self._optimizers

    def optimizers(self):
        if getattr(self, '_optimizers', None) is None:
            self.configure_optimizers()
        return self._optimizers

    def set_gradient_exchange(self, reducers):
        """data-parallel training: the three reducers of training.cubegan_reducers(self, self.optimizers()) (one exchange per backward pass)"""
        self._reducers = reducers

    def set_logger(self, fn):
        self._log_fn = fn

    def log_dict(self, d, **kw):
        fn = getattr(self, '_log_fn', None)
        if fn is not None:
            fn(d)

    def training_step(self, batch, batch_ids=None, rng=None):
        """cubegan.py:85-189: discriminator step, generator step (adversarial + feature matching + 45 x mel-L1), text step; returns the
        reference's dict (values as floats, read back from the device when first looked at: training.StepLosses)"""
        from .training import cubegan_training_step
        out = cubegan_training_step(self, batch, self.optimizers(), getattr(self, '_reducers', None), rng=rng)
        derived = lambda d: {'loss_v': d['loss_g'] + d['loss_d'], 'loss': d['loss_g'] + d['loss_d'] + d['loss_t']}
        if hasattr(out, 'also'):
            out.also(derived)      # (training.StepLosses: evaluated when the values have arrived, nothing waits here)
        else:
            out.update(derived(out))
        self.log_dict(out, prog_bar=True)
        return out

    def validation_step(self, batch, batch_ids=None, rng=None):
        """cubegan.py:191-273 (the quantity `.best` is selected on is loss_mel; the adversarial terms the reference only logs are not evaluated)"""
        from .training import cubegan_validation_step
        return cubegan_validation_step(self, batch, rng=rng)

    def validation_epoch_end(self, outputs) -> None:
        """cubegan.py:270-273"""
        self._val_loss = sum(x['loss_mel'] for x in outputs) / len(outputs)

    @torch.jit.ignore
    def save(self, path):
        torch.save(self.state_dict(), path)
        # the generator's split-precision calibration travels beside the checkpoint (<path>.scales.json): a reloaded model then
        # reproduces this handle's arithmetic without calibrating again
        rec = self._generator.export_scales()
        if rec is not None:
            import json
            json.dump(rec, open(path + '.scales.json', 'w'))

    @torch.jit.ignore
    def load(self, path):
        self.load_state_dict(torch.load(path, map_location='cpu'), strict=False)
        import json
        import os
        if os.path.exists(path + '.scales.json'):
            self._generator.import_scales(json.load(open(path + '.scales.json')))

    @staticmethod
    def _compute_lr(initial_lr, delta, step):
        return initial_lr / (1 + delta * step)

    def get_device(self):
        return self._languasito._get_device()
