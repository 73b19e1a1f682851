"""Independent sub-graphs of the training step on separate HIP streams (hifigan/disc_hip.py: the 5 + 3 sub-discriminators; hifigan/autograd.py:
the three ResBlock branches of a generator stage).  At the reference's batch of 16 crops a layer is a few hundred workgroups on 256 CUs (a
1024 x 1024 x 5 discriminator layer: ~320): launches of different sub-graphs fill each other's idle CUs, in the forward pass and — autograd replays
a node on the stream its forward ran on — in the backward pass.  TTSC_DISC_STREAMS=0 keeps everything on the current stream."""
import os

import torch

DISC_STREAMS = os.environ.get('TTSC_DISC_STREAMS', '1') != '0'
MAX_SIDE = int(os.environ.get('TTSC_SIDE_STREAMS_MAX', '64'))   # jobs beyond this many share streams round-robin
_SIDE = {}
_OFF_TAGS = set(v for v in os.environ.get('TTSC_STREAMS_OFF', '').split(',') if v)   # measurement switch: 'gen', 'mpd', 'msd'


_TEXT = {}
_EXCHANGE = {}
N_RESERVED = 8     # side streams taken together with the text stream at first use (5 period + 3 scale sub-discriminators)


def _reserve(dev):
    """All streams this package uses beside the caller's, taken from torch's pool in ONE go and in a fixed order — text stream, then the side streams.
    The HIP runtime multiplexes the streams of a process onto four hardware queues in creation order, so WHICH queue the text stream and each side stream
    share with the main stream depends on what was created before them: a Cubegan training step ran 61.5 ms in a fresh process and 69 ms after a small-batch
    generator forward had created two branch streams first (profiles/r06_branch_stream_queues.log).  With one reservation, and the generator's branch
    schedule borrowing two of these streams (ttsc_hifigan_set_branch_streams), the mapping no longer depends on the order in which the features of the
    package are first used."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _TEXT:
        d = torch.device('cuda', key)
        _TEXT[key] = torch.cuda.Stream(device=d)
        _SIDE[key] = [torch.cuda.Stream(device=d) for _ in range(N_RESERVED)]
        _EXCHANGE[key] = torch.cuda.Stream(device=d)
    return key


def exchange_stream(dev):
    """the ONE stream the gradient exchanges of a step are prepared and sent from (distributed.ArenaReducer: the three reducers of a Cubegan step share it)"""
    return _EXCHANGE[_reserve(dev)]


def text_stream(dev):
    """the stream of the text side of a training step (networks/training.py), normal priority"""
    return _TEXT[_reserve(dev)]


def _side_streams(dev, n):
    ss = _SIDE[_reserve(dev)]
    while len(ss) < n:
        ss.append(torch.cuda.Stream(device=dev))
    return ss[:n]


def branch_stream_handles(dev):
    """raw handles of the two side streams the generator's inference branch schedule borrows (hifigan/models.py)"""
    a, b = _side_streams(dev, 2)
    return a.cuda_stream, b.cuda_stream


def _tensors(nest):
    if torch.is_tensor(nest):
        yield nest
    elif isinstance(nest, (list, tuple)):
        for v in nest:
            yield from _tensors(v)


def fan_out(jobs, dev, inputs=(), tag=''):
    """run the thunks `jobs` (each returns a nest of tensors) on one side stream each; results are safe to use on the current stream afterwards.
    inputs: the tensors the jobs read that were allocated OUTSIDE (on the current stream).  They are marked as used by every side stream
    (`record_stream`): autograd keeps them as saved tensors and reads them again in the backward pass ON the side stream, then drops them — without
    the mark their memory returns to the current stream's pool at that moment and can be handed out and rewritten while the side stream's kernel
    is still reading it (round 4: with more than the runtime's four hardware queues really running side by side, two identical 5-step runs
    differed in 400 of 496 parameter tensors; tools/probes/train_determinism_poisoned.py)."""
    if not DISC_STREAMS or len(jobs) < 2 or tag in _OFF_TAGS:
        return [j() for j in jobs]
    main = torch.cuda.current_stream(dev)
    outs = []
    pool = _side_streams(dev, min(len(jobs), MAX_SIDE))
    for st in pool:
        st.wait_stream(main)
        for t in _tensors(list(inputs)):
            if t.is_cuda:
                t.record_stream(st)
    for i, job in enumerate(jobs):
        st = pool[i % len(pool)]
        with torch.cuda.stream(st):
            r = job()
        for t in _tensors(r):
            t.record_stream(main)      # allocated on the side stream, consumed (and freed) on the main one
        outs.append((st, r))
    for st in pool:
        main.wait_stream(st)
    return [r for _, r in outs]


def side_streams_of(dev):
    """the side streams created so far on `dev` (for code that must order its own stream behind all of them)"""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    return list(_SIDE.get(key, []))


def join_side_streams(dev, include_default=False):
    """make the CURRENT stream wait for everything queued so far on the side streams of `dev`.  For code that runs inside a backward pass and reads
    results of several sub-graphs — the gradient-exchange hooks of distributed.ArenaReducer: the host-side order of autograd nodes says nothing about
    the completion order of kernels on different streams.  include_default: also wait for the device's default stream — a hook can fire while the
    current stream IS a side stream (an AccumulateGrad node runs on the stream of the parameter's first use), and gradients of the same chunk may
    have been written on the main stream."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    cur = torch.cuda.current_stream(dev)
    for st in _SIDE.get(key, []):
        if st != cur:
            cur.wait_stream(st)
    if include_default:
        main = torch.cuda.default_stream(dev)
        if main != cur:
            cur.wait_stream(main)
