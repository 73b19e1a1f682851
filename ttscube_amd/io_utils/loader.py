"""Corpus sharding and background batch loading for the trainers (scripts/train_cubegan.py, scripts/train_vocoder.py).

The reference leaves both to Lightning's DistributedSampler + torch DataLoader workers (scripts/train_cubegan.py:100-126).  Here:
  * `rank_shard` — every rank gets EXACTLY ceil(n / world) item indices (wrap-padded, the DistributedSampler convention), so all
    ranks issue the same number of training steps and therefore the same number of gradient exchanges: a rank with one batch more
    than its peers would block forever in its extra reduce_scatter.
  * `BatchLoader` — items are read (wav / mgc / pitch decode) and collated by `num_workers` host threads one or two batches ahead
    of the GPU step instead of materialising the whole shard up front."""
import queue
import threading
from concurrent.futures import ThreadPoolExecutor


def rank_shard(n_items, rank, world):
    if n_items <= 0:
        return []
    per = -(-n_items // world)
    return [(rank + i * world) % n_items for i in range(per)]


def equal_batches(indices, batch_size):
    """consecutive batches of `indices`; with equal shard sizes every rank gets the same number of batches"""
    return [indices[s:s + batch_size] for s in range(0, len(indices), batch_size)]


def pin_batch(batch):
    """page-lock the tensors of a collated batch (on a machine with a HIP device), so that the training step's host-to-device copies are
    asynchronous: a pageable `.to(device)` blocks the host until the copy has run — 13 copies and ~4 ms of a 72 ms Cubegan step with the GPU idle
    (round 6, tools/probes/train_host_profile.py).  Done here, one batch ahead, on the loader's thread."""
    import torch
    if not (isinstance(batch, dict) and torch.cuda.is_available()):
        return batch
    for k, v in list(batch.items()):
        if torch.is_tensor(v) and v.device.type == 'cpu' and not v.is_pinned() and v.numel() > 0:
            try:
                batch[k] = v.pin_memory()
            except RuntimeError:
                return batch
    return batch


class BatchLoader:
    """Iterates collate([dataset[i] for i in batch]) over `batches` (lists of indices), prepared `depth` batches ahead by
    `num_workers` threads (numpy / soundfile release the GIL while decoding).  num_workers = 0 loads synchronously."""

    def __init__(self, dataset, batches, collate, num_workers=4, depth=2):
        self._ds, self._batches, self._collate = dataset, list(batches), collate
        self._nw, self._depth = max(0, int(num_workers)), max(1, int(depth))

    def __len__(self):
        return len(self._batches)

    def _load(self, pool, batch):
        items = list(pool.map(self._ds.__getitem__, batch)) if pool else [self._ds[i] for i in batch]
        return pin_batch(self._collate(items))

    def __iter__(self):
        if self._nw == 0:
            for b in self._batches:
                yield self._load(None, b)
            return
        q = queue.Queue(maxsize=self._depth)
        stop = threading.Event()

        def producer():
            try:
                with ThreadPoolExecutor(self._nw) as pool:
                    for b in self._batches:
                        if stop.is_set():
                            return
                        q.put(('ok', self._load(pool, b)))
                q.put(('end', None))
            except BaseException as e:   # surface loader errors in the training loop, never hang it
                q.put(('err', e))

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        try:
            while True:
                tag, val = q.get()
                if tag == 'end':
                    return
                if tag == 'err':
                    raise val
                yield val
        finally:
            stop.set()
            while th.is_alive():   # unblock a producer waiting on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    th.join(0.01)
