"""Data-parallel gradient exchange for the training step: one process per GPU, `torch.distributed`
(backend "nccl" == RCCL on ROCm, over xGMI), ONE exchange per backward pass as a few large flat fp32 buckets.

The reference has no explicit collective — it relies on Lightning's implicit DDP (train_cubegan.py:138-143).  Here the
exchange is explicit: gradients of a parameter group are packed into contiguous fp32 buckets and reduced with
reduce_scatter + all_gather (every GPU talks to all 7 peers at once over its point-to-point xGMI links, instead of a
ring all-reduce whose per-link traffic bounds it), launched asynchronously so that buckets overlap each other and the
optimizer's host-side work.  With world_size == 1 it is a no-op."""
import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class FlatBucketReducer:
    """Average the .grad of `params` across ranks.  bucket_mb: bucket size in MiB (large buckets: xGMI links are
    bandwidth- not latency-friendly; 64 MiB keeps ~6 buckets in flight for the 98 M-parameter Cubegan).

    use_reduce_scatter=True: reduce_scatter_tensor into a separate shard buffer + all_gather_into_tensor back (every GPU
    drives all of its xGMI links); False: one all_reduce per bucket.  Both run on nccl (RCCL) and gloo.
    A parameter that has no gradient on ANY rank keeps `.grad is None` (decided once, collectively, at the first reduce), so
    the optimizer treats it exactly as in a single-process run; one that has a gradient on some ranks gets the average
    with zeros for the others.  `force=True` runs the exchange even with a world of one process (tests)."""

    def __init__(self, params, bucket_mb=64, group=None, use_reduce_scatter=True, force=False):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.use_rs = use_reduce_scatter
        self.force = force
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)
        self._buckets = None
        self.bytes_exchanged = 0   # payload bytes handed to the collectives by the last reduce() (per rank)

    def _active(self):
        return dist.is_available() and dist.is_initialized() and (self.force or dist.get_world_size(self.group) > 1)

    def _build(self):
        world = dist.get_world_size(self.group)
        # which parameters take part: those with a gradient on at least one rank (one small collective, once)
        dev = self.params[0].device if self.params else torch.device('cpu')
        has = torch.tensor([1.0 if p.grad is not None else 0.0 for p in self.params], dtype=torch.float32, device=dev)
        if has.numel():
            dist.all_reduce(has, op=dist.ReduceOp.MAX, group=self.group)
        live = [p for p, h in zip(self.params, has.tolist()) if h > 0]
        buckets, cur, n = [], [], 0
        for p in live:
            if n + p.numel() > self.bucket_elems and cur:
                buckets.append(cur)
                cur, n = [], 0
            cur.append(p)
            n += p.numel()
        if cur:
            buckets.append(cur)
        self._buckets = []
        for ps in buckets:
            total = sum(p.numel() for p in ps)
            padded = (total + world - 1) // world * world
            flat = torch.zeros(padded, dtype=torch.float32, device=ps[0].device)
            shard = torch.empty(padded // world, dtype=torch.float32, device=ps[0].device)
            views, off = [], 0
            for p in ps:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            self._buckets.append((ps, flat, shard, views))

    @torch.no_grad()
    def reduce(self):
        """all-reduce (mean) of every participating parameter's .grad."""
        if not self._active():
            return
        if self._buckets is None:
            self._build()
        else:
            live = {id(p) for ps, _, _, _ in self._buckets for p in ps}
            for p in self.params:   # a parameter that got its first gradient after the buckets were laid out would never be exchanged
                if p.grad is not None and id(p) not in live:
                    raise RuntimeError('FlatBucketReducer: a parameter of shape %s received its first gradient after the first exchange; '
                                       'the replicas would diverge silently' % (tuple(p.shape),))
        world = dist.get_world_size(self.group)
        works = []
        self.bytes_exchanged = 0
        for ps, flat, shard, views in self._buckets:
            # pack: ONE multi-tensor copy per bucket (a per-parameter copy_ is a launch each: ~900 tensors x 3 exchanges per step)
            dst_l, src_l = [], []
            for p, v in zip(ps, views):
                if p.grad is None:
                    v.zero_()
                elif p.grad.data_ptr() != v.data_ptr():   # (a gradient that still aliases its view is already in place)
                    dst_l.append(v)
                    src_l.append(p.grad)
            if dst_l:
                torch._foreach_copy_(dst_l, src_l)
            flat.div_(world)
            if self.use_rs:
                w1 = dist.reduce_scatter_tensor(shard, flat, group=self.group, async_op=True)
            else:
                w1 = dist.all_reduce(flat, group=self.group, async_op=True)
            works.append(w1)
            self.bytes_exchanged += flat.numel() * 4 * (2 if self.use_rs else 1)
        gathers = []
        for w1, (ps, flat, shard, views) in zip(works, self._buckets):
            w1.wait()
            gathers.append(dist.all_gather_into_tensor(flat, shard, group=self.group, async_op=True) if self.use_rs else None)
        for w2, (ps, flat, shard, views) in zip(gathers, self._buckets):
            if w2 is not None:
                w2.wait()
            # unpack without copying: the averaged gradient of a parameter IS its slice of the bucket (the optimizer reads it there;
            # the next zero_grad(set_to_none=True) drops the alias, zero_grad(set_to_none=False) clears the slice)
            for p, v in zip(ps, views):
                p.grad = v


# measurement switches (tools/probes/train_hook_overhead.py): which stream prepares and sends an early chunk ('own' = a dedicated stream that waits for
# the default stream and all side streams; 'current' = round 3's behaviour), and whether anything leaves before reduce() at all
_EX_STREAM_MODE = __import__('os').environ.get('TTSC_EXCHANGE_STREAM', 'own')
_EX_STREAM_SHARED = _EX_STREAM_MODE != 'per_reducer'
if _EX_STREAM_MODE == 'per_reducer':
    _EX_STREAM_MODE = 'own'
_EARLY_OFF = __import__('os').environ.get('TTSC_EXCHANGE_EARLY', '1') == '0'


class ArenaReducer:
    """Gradient exchange of a `ttscube_amd.optim.FlatAdamW` group, overlapped with the backward pass that produces the gradients.

    The optimizer's gradient arena IS the exchange buffer: it is cut into `bucket_mb` chunks; every live parameter carries a
    post-accumulate-grad hook, and the moment the last parameter of a chunk has its gradient, that chunk's reduce_scatter is
    launched (async, on RCCL's stream) while autograd keeps differentiating the layers in front of it — what Lightning's DDP does
    per backward for the reference (scripts/train_cubegan.py:138-145), here with reduce_scatter + all_gather so that every GPU
    drives all 7 of its xGMI links.  `reduce()` after backward() launches whatever is still pending, then all_gathers.

    Issue order: every rank issues the chunks' collectives in ONE agreed order, whatever its own gradients do.  A parameter is
    live when ANY rank has a gradient for it, so a rank without that gradient never sees the chunk complete during backward(); if each rank
    launched chunks "when ready", RCCL would pair chunk k of one rank with chunk j of another (all chunks but the tail have the same size): wrong
    averages, no error.  So the order is fixed once — the order in which rank 0 saw the chunks complete during the first overlapped backward
    pass, broadcast to everybody — and a chunk leaves from a hook only when it is ready AND every chunk before it in that order has left; what is
    still pending leaves from `reduce()` in the same order.  A rank whose chunk never completes just launches the tail late.
    The first step has no arenas yet (liveness is decided from its gradients): `reduce()` builds them and exchanges un-overlapped; the second
    step records the order (un-overlapped as well); overlap starts with the third."""

    def __init__(self, opt, bucket_mb=64, group=None, use_reduce_scatter=True, force=False, overlap=True):
        self.opt, self.group, self.use_rs, self.force, self.overlap = opt, group, use_reduce_scatter, force, overlap
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)
        self.bytes_exchanged = 0
        self.launched_early = 0      # chunks whose reduce_scatter left from a gradient hook during the last backward pass
        self._chunks = None
        self._order = None           # agreed issue order of the chunks (list of chunk indices), None until recorded
        self._n_early = 0            # how many chunks at the head of _order may leave from gradient hooks (see _agree_on_order)
        self._seen = []              # recording pass: chunks in the order they completed on this rank
        self._next = 0               # position in _order of the next chunk to issue
        self._ex_stream = None       # the stream early chunks are prepared and sent from (see _launch)
        opt.on_build(self._on_build)

    def _active(self):
        return dist.is_available() and dist.is_initialized() and (self.force or dist.get_world_size(self.group) > 1)

    def _on_build(self, opt):
        world = dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1
        n = opt.g.numel()
        step = max(world, self.bucket_elems // world * world)
        self._chunks = []
        for s in range(0, n, step):
            e = min(n, s + step)
            assert (e - s) % world == 0 or e == n
            pad = (e - s + world - 1) // world * world
            # (the arena is padded to 64 elements per parameter; a ragged last chunk gets its own padded staging view)
            buf = opt.g[s:e] if pad == e - s else None
            self._chunks.append({'s': s, 'e': e, 'pad': pad, 'buf': buf, 'shard': torch.empty(pad // world, dtype=torch.float32, device=opt.g.device),
                                 'need': 0, 'left': 0, 'work': None})
        self._owners = []
        for i, o in zip(opt.live, opt.offsets):
            p = opt.params[i]
            cs = [k for k, c in enumerate(self._chunks) if c['s'] < o + p.numel() and o < c['e']]
            for k in cs:
                self._chunks[k]['need'] += 1
            self._owners.append(cs)
            if self.overlap:
                p.register_post_accumulate_grad_hook(self._make_hook(cs))
        self.arm()

    def _make_hook(self, cs):
        def hook(_p):
            if not self._armed:
                return
            for k in cs:
                c = self._chunks[k]
                c['left'] -= 1
                if c['left'] == 0:
                    if self._order is None:
                        self._seen.append(k)     # recording pass: nothing leaves early
                    else:
                        self._issue_ready()
        return hook

    def _issue_ready(self):
        """launch, in the agreed order, every chunk that is complete and whose predecessors have all left"""
        if _EARLY_OFF:   # (measurement switch TTSC_EXCHANGE_EARLY=0: the hooks count, nothing leaves before reduce())
            return
        while self._next < self._n_early:
            c = self._chunks[self._order[self._next]]
            if c['left'] != 0 or c['work'] is not None:
                break
            self._launch(c)
            self.launched_early += 1
            self._next += 1

    def _agree_on_order(self):
        """rank 0's completion order of the recording pass becomes everybody's issue order — restricted to the chunks that completed during
        backward() on EVERY rank (MIN over the ranks of a per-chunk flag).  The others (a parameter without a gradient on some rank keeps its
        chunk from completing there) go to the END of the order and are `late`: they never leave from a hook, on any rank, even where they do
        complete.  Why (round 6): several reducers of one step share a communicator (the Cubegan step arms the text-side reducer, then the
        discriminators'); a chunk that left early on one rank and from reduce() on another would sit on different sides of the OTHER reducer's
        collectives in the two ranks' launch sequences — mispaired operations.  With this rule every rank's sequence is: the common chunks in the
        agreed order (from hooks or, under skew, from reduce() — nothing else is launched in between), then the late chunks from reduce()."""
        n = len(self._chunks)
        mine = torch.zeros(n, dtype=torch.int64, device=self.opt.g.device)
        seen = list(dict.fromkeys(self._seen))
        if seen:
            mine[torch.tensor(seen, dtype=torch.int64, device=mine.device)] = 1
        dist.all_reduce(mine, op=dist.ReduceOp.MIN, group=self.group)
        common = set(int(k) for k in torch.nonzero(mine).flatten().tolist())
        first = [k for k in seen if k in common]
        order = first + [k for k in range(n) if k not in set(first)]
        t = torch.tensor([len(first)] + order, dtype=torch.int64, device=self.opt.g.device)
        dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        vals = [int(v) for v in t.tolist()]
        n_early, order = vals[0], vals[1:]
        if sorted(order) != list(range(n)) or not set(order[:n_early]) <= common:
            raise RuntimeError('ArenaReducer: the broadcast issue order is not a permutation of the chunks, or names an early chunk this rank '
                               'never saw complete (ranks built different arenas?)')
        self._order = order
        self._n_early = n_early      # chunks order[:n_early] may leave from hooks; the rest only from reduce()

    def arm(self):
        """call before a backward pass (right after zero_grad): chunk counters restart"""
        self._armed = self._chunks is not None and self._active() and self.overlap
        if self._chunks is not None:
            for c in self._chunks:
                c['left'], c['work'] = c['need'], None
        self.launched_early = 0
        self._next = 0
        self._seen = []
        # the stream this backward pass is started from (a caller may run the whole step under a non-default stream; the Cubegan step arms its
        # text-side reducer under the text stream): gradients produced there must be ordered before an early chunk's gather / div / send too
        self._arm_stream = torch.cuda.current_stream(self.opt.g.device) if (self.opt.built and self.opt.g.is_cuda) else None

    @torch.no_grad()
    def _launch(self, c):
        world = dist.get_world_size(self.group)
        g = self.opt.g
        ctx = None
        if g.is_cuda and self._armed:
            # Launched from a gradient hook, i.e. inside backward(): the parameters of this chunk may have received their gradients on different
            # streams (hifigan/streams.py runs independent sub-graphs on side streams; an AccumulateGrad node runs on the stream of the parameter's
            # first use, which may be the default stream or a side stream).  The chunk is therefore prepared and sent from a stream of its OWN
            # that waits for the current stream, the default stream and every side stream — none of THEM waits for anything, so the backward
            # pass keeps its multi-stream overlap (round 4: making the hook's current stream do the waiting cost 3.6 ms per step).
            from .hifigan.streams import join_side_streams, side_streams_of
            _mode = _EX_STREAM_MODE
            if _mode == 'own':
                if self._ex_stream is None:
                    # one exchange stream per device for every reducer, out of the package's reserved pool (a stream per reducer was three more
                    # streams on the runtime's four hardware queues: hifigan/streams.py::_reserve); TTSC_EXCHANGE_STREAM=per_reducer: round 5's
                    from .hifigan.streams import exchange_stream
                    self._ex_stream = exchange_stream(g.device) if _EX_STREAM_SHARED else torch.cuda.Stream(device=g.device)
                ex = self._ex_stream
                ex.wait_stream(torch.cuda.current_stream(g.device))
                ex.wait_stream(torch.cuda.default_stream(g.device))
                if getattr(self, '_arm_stream', None) is not None:
                    ex.wait_stream(self._arm_stream)
                for st in side_streams_of(g.device):
                    ex.wait_stream(st)
                ctx = torch.cuda.stream(ex)
                ctx.__enter__()
            elif _mode == 'current':      # (measurement: round 3's behaviour — the hook's own stream waits for the side streams)
                join_side_streams(g.device)
            elif _mode == 'current+default':
                join_side_streams(g.device, include_default=True)
        try:
            self.opt.gather(c['s'], c['e'])   # the chunk's gradients, from autograd's tensors into the arena (no-op for those already there)
            if c['buf'] is None:   # ragged tail: stage into a padded buffer
                if 'stage' not in c:
                    c['stage'] = torch.zeros(c['pad'], dtype=torch.float32, device=g.device)
                c['stage'][:c['e'] - c['s']].copy_(g[c['s']:c['e']])
                flat = c['stage']
            else:
                flat = c['buf']
            flat.div_(world)
            c['flat'] = flat
            if self.use_rs:
                c['work'] = dist.reduce_scatter_tensor(c['shard'], flat, group=self.group, async_op=True)
            else:
                c['work'] = dist.all_reduce(flat, group=self.group, async_op=True)
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)

    @torch.no_grad()
    def reduce(self):
        if not self._active():
            return
        recording = self._armed and self._order is None   # (decided before the arenas are built: the building step has recorded nothing)
        self.opt.ensure_built()
        self._armed = False
        self.bytes_exchanged = 0
        if recording:
            self._agree_on_order()
        for k in (self._order if self._order is not None else range(len(self._chunks))):   # pending chunks, in the agreed order
            c = self._chunks[k]
            if c['work'] is None:
                self._launch(c)
        gathers = []
        for c in self._chunks:
            c['work'].wait()
            gathers.append(dist.all_gather_into_tensor(c['flat'], c['shard'], group=self.group, async_op=True) if self.use_rs else None)
            self.bytes_exchanged += c['pad'] * 4 * (2 if self.use_rs else 1)
        for w, c in zip(gathers, self._chunks):
            if w is not None:
                w.wait()
            if c['buf'] is None:
                self.opt.g[c['s']:c['e']].copy_(c['flat'][:c['e'] - c['s']])
            c['work'] = None
        self.opt.grads_in_arena()      # `.grad` of every live parameter = its (averaged) slice of the arena


def broadcast_parameters(module, src=0, group=None):
    """Replicated parameters: every rank starts from rank `src`'s values."""
    if not is_dist():
        return
    ts = list(module.parameters()) + list(module.buffers())
    for t in ts:
        dist.broadcast(t.data, src=src, group=group)
    # written through `.data`: move the version counters, which the weight caches of the HIP layers key on (hip_layers.LSTMHip._sync, hifigan/wbank.py)
    torch.autograd.graph.increment_version(ts)
