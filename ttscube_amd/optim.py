"""AdamW over flat arenas — the optimizers of `Cubegan.configure_optimizers` (cube/networks/cubegan.py:275-311: three
torch.optim.AdamW(betas=(0.8, 0.99)) over ~900 parameter tensors) as ONE HIP kernel per group and step.

Layout: the live parameters of a group are re-pointed (``p.data``) into one contiguous fp32 arena; their gradients are gathered into a
second arena — the gradient-exchange bucket of `ttscube_amd.distributed.ArenaReducer` — chunk by chunk as they become ready (a few fused
copy launches per chunk; `zero_grad()` sets ``.grad = None`` so that autograd hands gradient tensors over without a `+=` launch each), and
``p.grad`` is the arena view again once the arena holds the step's gradients; the two moment estimates are two more arenas.
`step()` = `ttsc_adamw_step` over the four arenas (csrc/train_ops.hip).

"Live" = has a gradient on some rank after the first backward pass (decided once, collectively): a parameter nobody
differentiates keeps ``.grad is None`` and is never touched, exactly like torch.optim.AdamW skips it.  A parameter that turns up
with a gradient later raises (it would silently never be exchanged).

`state_dict()` / `load_state_dict()` speak torch.optim.AdamW's format ({'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}},
'param_groups': [...]}), so `<base>.opt.last` files (scripts/train_cubegan.py) stay interchangeable with torch's optimizer."""
import torch
import torch.distributed as dist

from . import _lib

_ALIGN = 64   # elements: every parameter starts on a 256-byte boundary of the arena
GATHER_GRADS = __import__('os').environ.get('TTSC_GRAD_GATHER', '1') != '0'


class FlatAdamW:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, group=None):
        self.params = [p for p in params]
        self.param_groups = [{'lr': lr, 'betas': tuple(betas), 'eps': eps, 'weight_decay': weight_decay, 'amsgrad': False,
                              'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                              'params': list(range(len(self.params)))}]
        self.group = group
        self.step_count = 0
        self.live = None          # indices into self.params, set by _build
        self.offsets = None       # arena offset of every live parameter
        self.p = self.g = self.m = self.v = None
        self._pending = None      # state loaded before the arenas exist
        self._hooks = []          # called once the arenas exist (the reducer registers its gradient hooks there)
        self._dirty = False       # gradients of the current step are still in autograd's own tensors (see zero_grad / gather)

    # ---- arenas ----------------------------------------------------------------------------------------------------------
    @property
    def built(self):
        return self.p is not None

    def _build(self):
        ps = self.params
        dev = ps[0].device   # (the layout itself is device-agnostic — the gloo tests build it on the CPU; step() needs the HIP kernel)
        has = torch.tensor([1.0 if (p.grad is not None and p.requires_grad) else 0.0 for p in ps], dtype=torch.float32, device=dev)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(has, op=dist.ReduceOp.MAX, group=self.group)   # live on ANY rank -> live everywhere (same arena layout)
        self.live = [i for i, h in enumerate(has.tolist()) if h > 0]
        offs, n = [], 0
        for i in self.live:
            offs.append(n)
            n += (ps[i].numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.offsets, self.numel = offs, n
        z = lambda: torch.zeros(max(n, _ALIGN), dtype=torch.float32, device=dev)
        self.p, self.g, self.m, self.v = z(), z(), z(), z()
        dst_p, src_p, dst_g, src_g = [], [], [], []
        for i, o in zip(self.live, offs):
            p = ps[i]
            if p.dtype != torch.float32:
                raise _lib.TTSCError('FlatAdamW: fp32 parameters only')
            vp, vg = self.p[o:o + p.numel()].view_as(p), self.g[o:o + p.numel()].view_as(p)
            dst_p.append(vp)
            src_p.append(p.data)
            if p.grad is not None:
                dst_g.append(vg)
                src_g.append(p.grad)
        with torch.no_grad():
            torch._foreach_copy_(dst_p, src_p)
            if dst_g:
                torch._foreach_copy_(dst_g, src_g)
        if dev.type == 'cuda':
            # the old parameter / gradient storages are dropped a few lines below and go back to the pool of the stream they were ALLOCATED on;
            # the copies above run on the CURRENT stream (the text side builds under its own stream): tell the allocator, or the next
            # allocation on the other stream may overwrite a source before its copy has executed (gather() does the same)
            cs = torch.cuda.current_stream(dev)
            for t_ in src_p + src_g:
                t_.record_stream(cs)
        self._gviews = []
        for i, o, vp in zip(self.live, offs, dst_p):
            p = ps[i]
            p.data = vp                                             # the parameter now IS its slice of the arena
            self._gviews.append(self.g[o:o + p.numel()].view_as(p))
            p.grad = self._gviews[-1]                               # this step's gradient was just copied there
        self._dirty = False
        self._seg_cache = {}
        if self._pending is not None:
            self._apply_state(self._pending)
            self._pending = None
        for h in self._hooks:
            h(self)

    def ensure_built(self):
        """lay the arenas out now (called after the FIRST backward pass, by the gradient exchange or by step())"""
        if not self.built:
            self._build()
        return self

    def on_build(self, fn):
        if self.built:
            fn(self)
        else:
            self._hooks.append(fn)

    def _check_no_stragglers(self):
        live = set(self.live)
        for i, p in enumerate(self.params):
            if i not in live and p.grad is not None and p.requires_grad:
                raise _lib.TTSCError('FlatAdamW: parameter #%d (%s) received its first gradient after the arenas were laid out; it would '
                                     'never be updated or exchanged. Make every differentiated path active in the first step.'
                                     % (i, tuple(p.shape)))

    # ---- torch.optim surface -------------------------------------------------------------------------------------------------
    def zero_grad(self, set_to_none=True):
        """Gradients are dropped, not cleared: with `.grad is None` autograd hands the gradient tensor of a backward pass over as it is (no
        kernel), where a standing view of the arena cost one `+=` launch per parameter and step (~900 per Cubegan step) plus the memset.
        `gather()` — called per exchange chunk by distributed.ArenaReducer, or by step() — copies them into the arena in a few fused launches."""
        if not self.built:
            for p in self.params:
                p.grad = None
            return
        if not GATHER_GRADS:      # (measurement switch: the round-3 scheme — standing arena views, one memset, autograd accumulates with `+=`)
            self.grads_in_arena()
            self.g.zero_()
            return
        for i in self.live:
            self.params[i].grad = None
        self._dirty = True

    def _segments(self, s, e):
        """live parameters overlapping the arena range [s, e), laid out once per range: parameters that lie wholly inside it as
        (parameter, its arena view), parameters cut by a range boundary as (parameter, first element, one past the last element, arena slice)"""
        key = (s, e)
        if key not in self._seg_cache:
            whole, parts = [], []
            for k, (i, o) in enumerate(zip(self.live, self.offsets)):
                n = self.params[i].numel()
                a, b = max(s, o), min(e, o + n)
                if a >= b:
                    continue
                if a == o and b == o + n:
                    whole.append((self.params[i], self._gviews[k]))
                else:
                    parts.append((self.params[i], a - o, b - o, self.g[a:b]))
            self._seg_cache[key] = (whole, parts)
        return self._seg_cache[key]

    @torch.no_grad()
    def gather(self, s=0, e=None):
        """bring the gradients autograd left in `.grad` into the arena range [s, e) (the whole arena by default); a parameter without a gradient
        contributes zeros.  Runs on the current stream."""
        whole, parts = self._segments(s, self.numel if e is None else e)
        dst, src, zero = [], [], []
        for p, view in whole:
            g = p.grad
            if g is view:
                continue
            if g is None:
                zero.append(view)
            else:
                dst.append(view)
                src.append(g)
        for p, a, b, d in parts:
            g = p.grad
            if g is None:
                zero.append(d)
            else:
                gv = g.reshape(-1)[a:b]
                if gv.data_ptr() != d.data_ptr():
                    dst.append(d)
                    src.append(gv)
        if dst:
            torch._foreach_copy_(dst, src)
            if self.g.is_cuda:
                # the gradient tensors were allocated by autograd on whatever stream produced them (side streams of hifigan/streams.py, the text
                # stream) and are dropped right after this copy (`grads_in_arena`): tell the allocator that THIS stream still reads them, or their
                # memory goes back to the producing stream's pool and can be rewritten before the copy has run
                cur = torch.cuda.current_stream(self.g.device)
                for t in src:
                    t.record_stream(cur)
        if zero:
            torch._foreach_zero_(zero)

    def grads_in_arena(self):
        """after the arena holds this step's (gathered, possibly averaged) gradients: `.grad` of every live parameter is its arena view again"""
        for k, i in enumerate(self.live):
            self.params[i].grad = self._gviews[k]
        self._dirty = False

    @torch.no_grad()
    def step(self, guard=None):
        """guard: a one-element device tensor (32-bit); the update is skipped ON THE DEVICE when it holds a non-zero value at execution time
        (ttsc_adamw_step_guarded: the word ttsc_split_status_collect left on this stream — the host does not wait to find out)"""
        if not self.built:
            self._build()
        if self.step_count < 4 or self.step_count % 64 == 0:     # (a Python loop over ~900 parameters: every step while the run is young, then sampled)
            self._check_no_stragglers()
        if self.p.device.type != 'cuda':
            raise _lib.TTSCError('FlatAdamW.step: parameters live on the CPU; move the model to a HIP device first (no CPU path)')
        if self._dirty:               # no exchange brought the gradients in: do it now.  `.grad` keeps pointing at autograd's tensors (the same
            self.gather()             # values as the arena now holds; re-pointing ~300 of them at their arena views cost 0.5 ms of host time per
            self._dirty = False       # optimizer and step with the GPU idle) until the next zero_grad() drops them
        self.step_count += 1
        pg = self.param_groups[0]
        with _lib.on_device(self.p.device):
            _lib.check(_lib.lib().ttsc_adamw_step_guarded(_lib.dev_ptr(self.p), _lib.dev_ptr(self.g), _lib.dev_ptr(self.m), _lib.dev_ptr(self.v),
                                                          self.numel, float(pg['lr']), float(pg['betas'][0]), float(pg['betas'][1]), float(pg['eps']),
                                                          float(pg['weight_decay']), self.step_count, guard.data_ptr() if guard is not None else None,
                                                          _lib.current_stream()), 'ttsc_adamw_step')
        # the kernel wrote through raw pointers: tell autograd / the weight caches of the inference handles (version counters)
        torch.autograd.graph.increment_version([self.params[i] for i in self.live])

    def state_dict(self):
        state = {}
        if self.built and self.step_count > 0:
            for i, o in zip(self.live, self.offsets):
                p = self.params[i]
                n = p.numel()
                state[i] = {'step': torch.tensor(float(self.step_count)), 'exp_avg': self.m[o:o + n].view_as(p).clone(),
                            'exp_avg_sq': self.v[o:o + n].view_as(p).clone()}
        elif self._pending is not None:
            state = self._pending['state']
        return {'state': state, 'param_groups': [dict(g) for g in self.param_groups]}

    def load_state_dict(self, sd):
        g = sd['param_groups'][0]
        for k in ('lr', 'betas', 'eps', 'weight_decay'):
            if k in g:
                self.param_groups[0][k] = tuple(g[k]) if k == 'betas' else g[k]
        if self.built:
            self._apply_state(sd)
        else:
            self._pending = sd

    def _apply_state(self, sd):
        st = sd['state']
        steps = []
        with torch.no_grad():
            for i, o in zip(self.live, self.offsets):
                e = st.get(i, st.get(str(i)))
                if e is None:
                    continue
                p = self.params[i]
                n = p.numel()
                self.m[o:o + n].view_as(p).copy_(e['exp_avg'])
                self.v[o:o + n].view_as(p).copy_(e['exp_avg_sq'])
                steps.append(int(float(e['step'])))
        if steps:
            self.step_count = max(steps)
