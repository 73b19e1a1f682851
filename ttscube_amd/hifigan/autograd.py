"""Native training path of the HiFi-GAN generator.

The reference trains the generator through torch autograd over torch convolutions (`Cubegan.training_step`,
cube/networks/cubegan.py:85-189; `Generator.forward` [EXTERNAL hifigan/models.py]).  Here every convolution of the
generator is one `torch.autograd.Function` whose three legs are hand-written HIP kernels behind the C ABI:

  forward   ttsc_conv1d_forward      implicit-GEMM fp32 MFMA, leaky-relu prologue / bias / residual epilogue fused
  dgrad     ttsc_conv1d_forward      the SAME kernel on the transposed+flipped weights; the epilogue multiplies by the
                                     leaky-relu derivative of the saved pre-activation (`gate_dev`) and adds the gradient
                                     of the residual branch
  wgrad     ttsc_conv_wgrad          fp32 MFMA correlation over all positions, split-K partials reduced in a fixed order

Weights live in torch parameters (weight-norm g, v stay torch leaves so the four AdamW optimizers of
cubegan.py:275-311 see the usual gradients); `ttsc_conv1d_set_weight_device` re-packs the MFMA fragments from the live
tensors on the stream every step.  ConvTranspose1d backward runs on the phase-de-interleaved output gradient
dyP[b, r*Co+co, q] = dy[b, co, q*u + r], which turns both legs into stride-1 problems of the same kernels.
Arithmetic: dense stride-1 convolutions run forward and data gradient on the split-precision kernel (`ttsc_conv_train`, csrc/conv_train.hip:
fp16 hi/lo x 3 products on MFMA with fp32 accumulation; the fp16 ranges are set per launch from device-side maxima of the tensor and of
the weights, so gradients of any magnitude are safe; on one forward graph its data gradients agree with the exact kernel's to 4e-7).
Dense weight gradients run on the split-precision 128 x 64 tile kernel (`ttsc_conv_wgrad_split`), grouped layers and 1-channel layers on the split
convolution too; ConvTranspose1d, grouped weight gradients and weight gradients with < 64 rows stay on the exact-fp32 MFMA kernels."""
import ctypes as C
import os

import numpy as np
import torch

from .. import _lib
from ..hip_layers import Conv1dHip
from .wbank import AmaxPool, range_of, tag_range


class TrainConv:
    """Per-layer state of the native training path: forward handle, data-gradient handle, index maps."""

    def __init__(self, Cin, Cout, K, stride=1, padding=0, dilation=1, transposed=False, groups=1):
        self.Cin, self.Cout, self.K = Cin, Cout, K
        self.stride, self.padding, self.dilation, self.transposed = stride, padding, dilation, transposed
        self.groups = groups
        self.fwd = Conv1dHip(Cin, Cout, K, stride=stride, padding=padding, dilation=dilation, transposed=transposed, groups=groups)
        self._dgrad = None
        self._maps = None
        self._bg_ws = {}

    # ---- Conv1d ----------------------------------------------------------------------------------------------------
    def dgrad_handle(self):
        if self._dgrad is None:
            if not self.transposed:
                pd = self.dilation * (self.K - 1) - self.padding
                if pd < 0:
                    raise _lib.TTSCError('TrainConv: padding larger than the receptive field is not supported')
                self._dgrad = Conv1dHip(self.Cout, self.Cin, self.K, padding=pd, dilation=self.dilation, groups=self.groups)
            else:
                self._dgrad = Conv1dHip(self.stride * self.Cout, self.Cin, self.taps_t()[2], padding=0)
        return self._dgrad

    # ---- ConvTranspose1d: phase decomposition ------------------------------------------------------------------------
    def taps_t(self):
        u, p, K = self.stride, self.padding, self.K
        m_lo = (0 - p) // u
        m_hi = (K - 1 - p) // u
        return m_lo, m_hi, m_hi - m_lo + 1

    def maps_t(self, device):
        """gather maps (device long tensors): w[Ci,Co,K] -> W'[Ci, u*Co, M]  and  G[u*Co, Ci, M] -> dW[Ci,Co,K]."""
        if self._maps is None:
            u, p, K, Ci, Co = self.stride, self.padding, self.K, self.Cin, self.Cout
            m_lo, _, M = self.taps_t()
            ci = np.arange(Ci)[:, None, None, None]
            r = np.arange(u)[None, :, None, None]
            co = np.arange(Co)[None, None, :, None]
            j = np.arange(M)[None, None, None, :]
            k = (m_lo + j) * u + r + p
            src = (ci * Co + co) * K + k
            src = np.where((k >= 0) & (k < K), src, Ci * Co * K)          # last slot = appended zero
            w_map = np.broadcast_to(src, (Ci, u, Co, M)).reshape(Ci, u * Co, M)
            ci = np.arange(Ci)[:, None, None]
            co = np.arange(Co)[None, :, None]
            k = np.arange(K)[None, None, :]
            r = (k - p) % u
            jj = (k - p - r) // u - m_lo
            g_map = ((r * Co + co) * Ci + ci) * M + jj
            self._maps = (torch.from_numpy(np.array(w_map)).long().to(device),
                          torch.from_numpy(np.array(np.broadcast_to(g_map, (Ci, Co, K)))).long().to(device))
        return self._maps

    def deinterleave(self, dy, Lin):
        """dy [B,Co,Lo] -> dyP [B, u*Co, Lin+M-1] with dyP[b, r*Co+co, i] = dy[b, co, (i+m_lo)*u + r] (zero outside)."""
        u = self.stride
        m_lo, _, M = self.taps_t()
        B, Co, Lo = dy.shape
        Lq = -(-Lo // u)
        if Lq * u != Lo:
            dy = torch.nn.functional.pad(dy, (0, Lq * u - Lo))
        ph = dy.view(B, Co, Lq, u).permute(0, 3, 1, 2).reshape(B, u * Co, Lq)
        out = torch.zeros((B, u * Co, Lin + M - 1), dtype=dy.dtype, device=dy.device)
        # out index i <-> q = i + m_lo
        q0, q1 = max(0, m_lo), min(Lq, Lin + M - 1 + m_lo)
        if q1 > q0:
            out[:, :, q0 - m_lo:q1 - m_lo] = ph[:, :, q0:q1]
        return out


def _bias_grad(tc, dy):
    """db[c] = sum_{b,t} dy[b,c,t] — one launch (ttsc_bias_grad); the ticket workspace is kept per layer and shape."""
    B, Cc, L = dy.shape
    Lb = _lib.lib()
    key = (B, Cc, L, dy.device)
    ws = tc._bg_ws.get(key)
    fresh = ws is None
    nbytes = int(Lb.ttsc_bias_grad_workspace_bytes(B, Cc, L))
    if fresh:
        ws = torch.zeros((nbytes + 3) // 4, dtype=torch.float32, device=dy.device)
        tc._bg_ws = {key: ws}
    db = torch.empty(Cc, dtype=torch.float32, device=dy.device)
    with _lib.on_device(dy.device):
        _lib.check(Lb.ttsc_bias_grad(_lib.dev_ptr(dy), _lib.dev_ptr(db), B, Cc, L, _lib.dev_ptr(ws), nbytes, int(fresh),
                                     _lib.current_stream()), 'ttsc_bias_grad')
    return db


def _wgrad(P, Q, A, Bc, J, base, step, q_scale, q_slope, groups=1, amax=None, p_measured=False, pooled=False, amax_q=None, amax_p=None, db=None):
    """G [A, Bc / groups, J]: weight gradient of a (grouped) Conv1d, torch layout.  amax: the layer's range words [max |Q|, max |w|, max |P|] from
    the forward launch (max |Q| valid; max |P| valid when `p_measured`), None = measured here.  amax_q / amax_p: the two words as separate
    one-element tensors instead (banked layers: they may belong to the launches that PRODUCED Q and P).  db: a list; when the split-precision
    kernel takes the layer, the bias gradient sum_{n,t} P[n,a,t] rides in its two launches and is appended to it (else the list stays empty)"""
    Bg = Bc // groups
    G = torch.empty((A, Bg, J), dtype=torch.float32, device=P.device)
    N, _, LP = P.shape
    LQ = Q.shape[2]
    L = _lib.lib()
    if SPLIT_TRAIN and groups == 1 and L.ttsc_conv_wgrad_split_supported(A, Bc, J, step):   # dense layer: fp16 hi/lo x 3 on MFMA, 128 x 64 tiles
        nbytes = int(L.ttsc_conv_wgrad_split_workspace_bytes(N, A, Bc, LP, J))
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=P.device)
        with _lib.on_device(P.device):
            if amax_q is not None:
                aq, ap, have = _lib.dev_ptr(amax_q), _lib.dev_ptr(amax_p), True
            else:
                aq = _lib.dev_ptr(amax[0:1]) if amax is not None else None
                ap = _lib.dev_ptr(amax[2:3]) if amax is not None else None
                have = amax is not None
            dbt = torch.empty(A, dtype=torch.float32, device=P.device) if db is not None else None
            _lib.check(L.ttsc_conv_wgrad_split_bias(_lib.dev_ptr(P), _lib.dev_ptr(Q), _lib.dev_ptr(G), _lib.dev_ptr(dbt) if dbt is not None else None,
                                                    N, A, Bc, LP, LQ, J, base, step, q_scale, q_slope, aq, ap,
                                                    0 if not have else ((0 if p_measured else 2) | (4 if pooled else 0)), _lib.dev_ptr(ws), nbytes,
                                                    _lib.current_stream()), 'ttsc_conv_wgrad_split')
            if dbt is not None:
                db.append(dbt)
        return G
    if SPLIT_TRAIN and GROUPED_SPLIT and groups > 1 and (A // groups > 32 or GROUPED_SPLIT_ALL) and L.ttsc_conv_wgrad_split_grouped_supported(A, Bg, groups, J, step):
        # grouped layer (MSD's k = 41 convolutions): every group a small dense problem on the split-precision tile kernel (round 6).  Only groups of more
        # than 32 rows (two MFMA row tiles share every column fragment): with ONE row tile the kernel is bound by its LDS fragment reads (two reads per
        # three matrix instructions) — 179 us per launch of the 128 -> 128 layer against 154 + 174 us for all 21 taps on the exact kernel, which runs
        # these shapes at 0.64 of the fp32 matrix peak (profiles/r06_grouped_wgrad_ab.log)
        nbytes = int(L.ttsc_conv_wgrad_split_grouped_workspace_bytes(N, A, Bg, groups, LP, J))
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=P.device)
        with _lib.on_device(P.device):
            if amax_q is not None:
                aq, ap, have = _lib.dev_ptr(amax_q), _lib.dev_ptr(amax_p), True
            else:
                aq = _lib.dev_ptr(amax[0:1]) if amax is not None else None
                ap = _lib.dev_ptr(amax[2:3]) if amax is not None else None
                have = amax is not None
            _lib.check(L.ttsc_conv_wgrad_split_grouped(_lib.dev_ptr(P), _lib.dev_ptr(Q), _lib.dev_ptr(G), N, A, Bg, groups, LP, LQ, J, base, step, q_scale,
                                                       q_slope, aq, ap, 0 if not have else ((0 if p_measured else 2) | (4 if pooled else 0)),
                                                       _lib.dev_ptr(ws), nbytes, _lib.current_stream()), 'ttsc_conv_wgrad_split_grouped')
        return G
    nbytes = int(L.ttsc_conv_wgrad_workspace_bytes(N, A, Bg, LP, J))
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=P.device)
    with _lib.on_device(P.device):
        _lib.check(L.ttsc_conv_wgrad_grouped(_lib.dev_ptr(P), _lib.dev_ptr(Q), _lib.dev_ptr(G), N, A, Bg, groups, LP, LQ, J, base, step,
                                             q_scale, q_slope, _lib.dev_ptr(ws), nbytes, _lib.current_stream()), 'ttsc_conv_wgrad')
    return G


# Dense stride-1 convolutions run forward and data gradient on the split-precision kernel (csrc/conv_train.hip: fp16 hi/lo x 3 on MFMA, ranges
# measured on the device per launch, batch folded into the tile columns); TTSC_TRAIN_SPLIT=0 keeps everything on the exact-fp32 kernel.
SPLIT_TRAIN = os.environ.get('TTSC_TRAIN_SPLIT', '1') != '0'
GROUPED_SPLIT = os.environ.get('TTSC_TRAIN_GROUPED_SPLIT', '1') != '0'   # (measurement switch: 0 = grouped weight gradients on the exact fp32 kernel, round 5's path)
GROUPED_SPLIT_ALL = os.environ.get('TTSC_TRAIN_GROUPED_SPLIT', '1') == '2'   # (2 = also groups of <= 32 rows: slower, see _wgrad)


def _split_ok(Cin, Cout, K, dilation, groups=1):
    return SPLIT_TRAIN and bool(_lib.lib().ttsc_conv_train_supported(Cin, Cout, K, dilation, groups))


def _conv_split(x, w, b, resid, gate, Cin, Cout, K, padding, dilation, flip, in_scale=1.0, in_slope=1.0, out_scale=1.0, gate_slope=1.0, groups=1,
                amax_x=None, amax_w=None, measure=3):
    """one launch group of ttsc_conv_train (range words + weight split + convolution); x [B,Cin,Lin] -> [B,Cout,Lout].
    amax_x / amax_w: one-element fp32 device tensors holding (or, per `measure` bit 0 / 1, receiving) max |x| / max |w|; None = internal"""
    L = _lib.lib()
    B, _, Lin = x.shape
    Lout = Lin + 2 * padding - dilation * (K - 1)
    y = torch.empty((B, Cout, Lout), dtype=torch.float32, device=x.device)
    nbytes = int(L.ttsc_conv_train_workspace_bytes(Cin, Cout, K, groups))
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    ptr = lambda t: _lib.dev_ptr(t) if t is not None else None
    with _lib.on_device(x.device):
        _lib.check(L.ttsc_conv_train(ptr(x), ptr(w), ptr(b), ptr(resid), ptr(gate), ptr(y), B, Cin, Cout, K, Lin, padding, dilation, groups, int(flip),
                                     float(in_scale), float(in_slope), float(out_scale), float(gate_slope), ptr(amax_x), ptr(amax_w), int(measure),
                                     ptr(ws), nbytes, _lib.current_stream()), 'ttsc_conv_train')
    return y


def _conv_packed(x, frag, amax_w, b, resid, gate, Cin, Cout, K, padding, dilation, amax_x, measure, in_scale=1.0, in_slope=1.0, out_scale=1.0,
                 gate_slope=1.0, groups=1, amax_y=None):
    """ttsc_conv_train_packed: the same convolution on fragments a WeightBank prepared (wbank.py) — the range reduction of x (unless amax_x already
    holds it) and the convolution, nothing else.  measure: bit 0 = reduce max |x| into amax_x now, bit 2 = amax_x is a pre-zeroed pooled word;
    amax_y: zeroed word that receives max |y| from the convolution's epilogue"""
    L = _lib.lib()
    B, _, Lin = x.shape
    Lout = Lin + 2 * padding - dilation * (K - 1)
    y = torch.empty((B, Cout, Lout), dtype=torch.float32, device=x.device)
    ptr = lambda t: _lib.dev_ptr(t) if t is not None else None
    with _lib.on_device(x.device):
        _lib.check(L.ttsc_conv_train_packed(ptr(x), ptr(frag), ptr(b), ptr(resid), ptr(gate), ptr(y), B, Cin, Cout, K, Lin, padding, dilation, groups,
                                            float(in_scale), float(in_slope), float(out_scale), float(gate_slope), ptr(amax_x), ptr(amax_w), int(measure),
                                            ptr(amax_y), _lib.current_stream()), 'ttsc_conv_train_packed')
    return y


USE_BANK = os.environ.get('TTSC_TRAIN_WBANK', '1') != '0'      # (measurement switch: 0 = per-launch weight preparation, round 5's path)


class HipConvFn(torch.autograd.Function):
    """y = conv(leaky_relu(in_scale * x, in_slope); w) + b [+ resid]  with HIP forward / dgrad / wgrad."""

    @staticmethod
    def forward(ctx, x, w, b, resid, tc, in_scale, in_slope, pack=None, words=None, amax_in=None):
        x = x.contiguous()
        ctx.pack = pack
        if pack is not None:
            # weight fragments of both orders and max |w| prepared for the whole module by a WeightBank; the range words come from the step's
            # pre-zeroed pool (one memset per step instead of one per launch): max |x| from the launch that produced x when it left one (amax_in),
            # else reduced here into words[0]; this launch leaves max |y| in words[1] for whoever reads y next
            ctx.words = words
            ctx.amax_x = amax_in if amax_in is not None else words[0:1]
            y = _conv_packed(x, pack[0], pack[2], b.detach().contiguous() if b is not None else None, resid.contiguous() if resid is not None else None,
                             None, tc.Cin, tc.Cout, tc.K, tc.padding, tc.dilation, ctx.amax_x, 4 if amax_in is not None else 5, in_scale=in_scale,
                             in_slope=in_slope, groups=tc.groups, amax_y=words[1:2])
            ctx.save_for_backward(x)
            ctx.tc, ctx.in_scale, ctx.in_slope = tc, in_scale, in_slope
            ctx.has_b, ctx.has_r = b is not None, resid is not None
            return y
        wd = w.detach().contiguous()
        if not tc.transposed and tc.stride == 1 and _split_ok(tc.Cin, tc.Cout, tc.K, tc.dilation, tc.groups):
            # range words of this layer for this step: [max |x|, max |w|, max |dy|] — each tensor is reduced once, by the first launch that needs it; the
            # words come zeroed from the step's pool (no memset launch per reduction)
            ctx.amax = AmaxPool.of(x.device).take()
            y = _conv_split(x, wd, b.detach().contiguous() if b is not None else None, resid.contiguous() if resid is not None else None, None,
                            tc.Cin, tc.Cout, tc.K, tc.padding, tc.dilation, 0, in_scale=in_scale, in_slope=in_slope, groups=tc.groups,
                            amax_x=ctx.amax[0:1], amax_w=ctx.amax[1:2], measure=7)
        else:
            tc.fwd.set_weight_device(wd, b.detach() if b is not None else None)
            y = tc.fwd(x, resid=resid, in_scale=in_scale, in_slope=in_slope)
        ctx.save_for_backward(x, wd)
        if not hasattr(ctx, 'amax'):
            ctx.amax = None
        ctx.tc, ctx.in_scale, ctx.in_slope = tc, in_scale, in_slope
        ctx.has_b, ctx.has_r = b is not None, resid is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        tc, sc, sl = ctx.tc, ctx.in_scale, ctx.in_slope
        dy = dy.contiguous()
        if ctx.pack is not None:
            (x,) = ctx.saved_tensors
            dx = dw = db = None
            words = ctx.words
            # max |dy|: left by the launch that produced dy (the next layer's data gradient) unless autograd has added another gradient into it since
            adv = range_of(dy)
            ap = adv if adv is not None else words[2:3]
            dy_measured = adv is not None
            if ctx.needs_input_grad[0]:
                pd = tc.dilation * (tc.K - 1) - tc.padding
                dx = _conv_packed(dy, ctx.pack[1], ctx.pack[2], None, None, x if sl != 1.0 else None, tc.Cout, tc.Cin, tc.K, pd, tc.dilation, ap,
                                  4 if dy_measured else 5, out_scale=sc, gate_slope=sl, groups=tc.groups, amax_y=words[3:4])
                tag_range(dx, words[3:4], AmaxPool.of(dx.device))
                dy_measured = True
            want_db = ctx.has_b and ctx.needs_input_grad[2]
            if ctx.needs_input_grad[1]:
                dbl = [] if want_db else None
                dw = _wgrad(dy, x, tc.Cout, tc.Cin, tc.K, -tc.padding, tc.dilation, sc, sl, tc.groups, p_measured=dy_measured, pooled=True,
                            amax_q=ctx.amax_x, amax_p=ap, db=dbl)
                if dbl:
                    db = dbl[0]
            if want_db and db is None:
                db = _bias_grad(tc, dy)
            dr = dy if (ctx.has_r and ctx.needs_input_grad[3]) else None
            return dx, dw, db, dr, None, None, None, None, None, None
        x, w = ctx.saved_tensors
        B, _, Lin = x.shape
        dx = dw = db = None
        dy_measured = False
        if not tc.transposed:
            if ctx.needs_input_grad[0]:
                pd = tc.dilation * (tc.K - 1) - tc.padding
                if tc.stride == 1 and pd >= 0 and _split_ok(tc.Cout, tc.Cin, tc.K, tc.dilation, tc.groups):
                    am = ctx.amax
                    dx = _conv_split(dy, w, None, None, x if sl != 1.0 else None, tc.Cout, tc.Cin, tc.K, pd, tc.dilation, 1, out_scale=sc,
                                     gate_slope=sl, groups=tc.groups, amax_x=am[2:3] if am is not None else None,
                                     amax_w=am[1:2] if am is not None else None, measure=5 if am is not None else 1)
                    dy_measured = am is not None
                else:
                    h = tc.dgrad_handle()
                    h.set_weight_device_dgrad(w)
                    dx = h(dy, out_scale=sc, gate=x if sl != 1.0 else None, gate_slope=sl)
            if ctx.needs_input_grad[1]:
                dw = _wgrad(dy, x, tc.Cout, tc.Cin, tc.K, -tc.padding, tc.dilation, sc, sl, tc.groups, amax=ctx.amax, p_measured=dy_measured,
                            pooled=ctx.amax is not None)
        else:
            m_lo, _, M = tc.taps_t()
            dyp = tc.deinterleave(dy, Lin)
            w_map, g_map = tc.maps_t(dy.device)
            if ctx.needs_input_grad[0]:
                wz = torch.cat([w.reshape(-1), w.new_zeros(1)])
                wd = wz[w_map].contiguous()                      # [Ci, u*Co, M]: the data gradient is a dense stride-1 convolution of dyP
                if _split_ok(tc.stride * tc.Cout, tc.Cin, M, 1):
                    dx = _conv_split(dyp, wd, None, None, x if sl != 1.0 else None, tc.stride * tc.Cout, tc.Cin, M, 0, 1, 0, out_scale=sc, gate_slope=sl)
                else:
                    h = tc.dgrad_handle()
                    h.set_weight_device(wd)
                    dx = h(dyp, out_scale=sc, gate=x if sl != 1.0 else None, gate_slope=sl)
            if ctx.needs_input_grad[1]:
                G = _wgrad(dyp, x, tc.stride * tc.Cout, tc.Cin, M, 0, -1, sc, sl)
                dw = G.reshape(-1)[g_map]
        if ctx.has_b and ctx.needs_input_grad[2]:
            db = _bias_grad(tc, dy)
        dr = dy if (ctx.has_r and ctx.needs_input_grad[3]) else None
        return dx, dw, db, dr, None, None, None, None, None, None


def hip_conv(tc, x, w, b=None, resid=None, in_scale=1.0, in_slope=1.0):
    """w from WeightBank.weight(i) carries the prepared fragments (`_ttsc_pack`); any other weight tensor is prepared per launch"""
    pack = getattr(w, '_ttsc_pack', None)
    if pack is None:
        return HipConvFn.apply(x, w, b, resid, tc, float(in_scale), float(in_slope), None, None, None)
    if not (SPLIT_TRAIN and not tc.transposed and tc.stride == 1):
        raise _lib.TTSCError('hip_conv: a banked weight needs the split-precision stride-1 path')
    pool = AmaxPool.of(x.device)
    words = pool.take()
    y = HipConvFn.apply(x, w, b, resid, tc, float(in_scale), float(in_slope), pack, words, range_of(x) if x.is_contiguous() else None)
    return tag_range(y, words[1:2], pool)


class HipWeightNormFn(torch.autograd.Function):
    """w = g * v / ||v||  (weight_norm dim=0) as one kernel forward, one backward."""

    @staticmethod
    def forward(ctx, v, g):
        v = v.contiguous()
        g = g.contiguous()
        R = v.shape[0]
        Cc = v.numel() // R
        w = torch.empty_like(v)
        n = torch.empty(R, dtype=torch.float32, device=v.device)
        with _lib.on_device(v.device):
            _lib.check(_lib.lib().ttsc_weight_norm_forward(_lib.dev_ptr(v), _lib.dev_ptr(g), _lib.dev_ptr(w), _lib.dev_ptr(n), R, Cc,
                                                           _lib.current_stream()), 'ttsc_weight_norm_forward')
        ctx.save_for_backward(v, g, n)
        return w

    @staticmethod
    def backward(ctx, dw):
        v, g, n = ctx.saved_tensors
        dw = dw.contiguous()
        R = v.shape[0]
        Cc = v.numel() // R
        dv = torch.empty_like(v)
        dg = torch.empty_like(g)
        with _lib.on_device(v.device):
            _lib.check(_lib.lib().ttsc_weight_norm_backward(_lib.dev_ptr(dw), _lib.dev_ptr(v), _lib.dev_ptr(g), _lib.dev_ptr(n),
                                                            _lib.dev_ptr(dv), _lib.dev_ptr(dg), R, Cc, _lib.current_stream()),
                       'ttsc_weight_norm_backward')
        return dv, dg


def _wn(l):
    """live weight-norm: w = g * v / ||v|| (so that gradients reach weight_g and weight_v)."""
    if hasattr(l, 'weight'):
        return l.weight
    return HipWeightNormFn.apply(l.weight_v, l.weight_g)


def _train_convs(gen):
    tcs = getattr(gen, '_train_convs_cache', None)
    if tcs is not None:
        return tcs
    from .models import ResBlock1
    h = gen.h
    tcs = {}
    c0 = h['upsample_initial_channel']
    tcs['conv_pre'] = TrainConv(h.get('num_mels', 80), c0, 7, padding=3)
    nk = gen.num_kernels
    ch = c0
    for i, (u, k) in enumerate(zip(h['upsample_rates'], h['upsample_kernel_sizes'])):
        tcs['ups.%d' % i] = TrainConv(ch, ch // 2, k, stride=u, padding=(k - u) // 2, transposed=True)
        ch //= 2
        for j in range(nk):
            rb = gen.resblocks[i * nk + j]
            kr, ds = h['resblock_kernel_sizes'][j], h['resblock_dilation_sizes'][j]
            for m, d in enumerate(ds):
                if isinstance(rb, ResBlock1):
                    tcs['rb.%d.c1.%d' % (i * nk + j, m)] = TrainConv(ch, ch, kr, padding=d * (kr - 1) // 2, dilation=d)
                    tcs['rb.%d.c2.%d' % (i * nk + j, m)] = TrainConv(ch, ch, kr, padding=(kr - 1) // 2)
                else:
                    tcs['rb.%d.c.%d' % (i * nk + j, m)] = TrainConv(ch, ch, kr, padding=d * (kr - 1) // 2, dilation=d)
    tcs['conv_post'] = TrainConv(ch, 1, 7, padding=3)
    object.__setattr__(gen, '_train_convs_cache', tcs)
    return tcs


def _generator_weights(gen, T):
    """-> weight(layer): the effective weights of the generator's stride-1 convolutions out of ONE WeightBank.prepare() (three launches for all 74
    layers and both operand orders); ConvTranspose1d layers and anything the split path does not take keep the per-layer weight norm"""
    if not (USE_BANK and SPLIT_TRAIN):
        return _wn
    from .models import ResBlock1
    bank = getattr(gen, '_wbank', None)
    if bank is None:
        from .wbank import WeightBank
        nk = gen.num_kernels
        layers = [('conv_pre', gen.conv_pre), ('conv_post', gen.conv_post)]
        for n, rb in enumerate(gen.resblocks):
            if isinstance(rb, ResBlock1):
                for m, (c1, c2) in enumerate(zip(rb.convs1, rb.convs2)):
                    layers += [('rb.%d.c1.%d' % (n, m), c1), ('rb.%d.c2.%d' % (n, m), c2)]
            else:
                layers += [('rb.%d.c.%d' % (n, m), c) for m, c in enumerate(rb.convs)]
        specs, index = [], {}
        for name, l in layers:
            tc = T[name]
            if hasattr(l, 'weight_g') and _split_ok(tc.Cin, tc.Cout, tc.K, tc.dilation) and _split_ok(tc.Cout, tc.Cin, tc.K, tc.dilation):
                index[id(l)] = len(specs)
                specs.append((l, tc.Cin, tc.Cout, tc.K, 1, 1))
        bank = WeightBank(specs) if specs else False
        object.__setattr__(gen, '_wbank', bank)
        object.__setattr__(gen, '_wbank_index', index)
    if bank is False:
        return _wn
    bank.prepare()
    index = gen._wbank_index

    def weight(l):
        i = index.get(id(l))
        return bank.weight(i) if i is not None else _wn(l)
    return weight


class _BranchSum(torch.autograd.Function):
    """r0 + r1 + ... (left to right) for branches that ran on DIFFERENT streams.  A plain `+` hands the SAME gradient tensor to every branch; each
    branch's last node (a convolution with a residual input) passes it on unchanged as the residual's gradient, and autograd then accumulates the
    other contribution INTO it in place once it holds the only reference — while kernels of the other branches, on their own streams, may still be
    reading it.  (Round 4, found with more than the runtime's four hardware queues running side by side: two identical 5-step runs differed in 400
    of 496 parameter tensors.)  Every branch but the first therefore gets a copy of its own."""

    @staticmethod
    def forward(ctx, *rs):
        ctx.n = len(rs)
        xs = rs[0]
        for r in rs[1:]:
            xs = xs + r
        return xs

    @staticmethod
    def backward(ctx, g):
        return (g,) + tuple(g.clone() for _ in range(ctx.n - 1))


def generator_forward_with_grad(gen, x):
    """Differentiable HiFi-GAN generator forward on the HIP kernels (same math as `ttsc_hifigan_forward`)."""
    from .models import ResBlock1
    from .streams import fan_out
    if not x.is_cuda:
        raise _lib.TTSCError('generator training needs a HIP device; no CPU path')
    h = gen.h
    T = _train_convs(gen)
    nk = gen.num_kernels
    _wn = _generator_weights(gen, T)           # (shadows the per-layer weight-norm launch: banked layers come prepared)
    x = hip_conv(T['conv_pre'], x.float(), _wn(gen.conv_pre), gen.conv_pre.bias)
    for i in range(len(h['upsample_rates'])):
        x = hip_conv(T['ups.%d' % i], x, _wn(gen.ups[i]), gen.ups[i].bias, in_scale=(1.0 / nk) if i > 0 else 1.0, in_slope=0.1)
        def branch(j, x=x, i=i):
            rb = gen.resblocks[i * nk + j]
            r = x
            if isinstance(rb, ResBlock1):
                for m, (c1, c2) in enumerate(zip(rb.convs1, rb.convs2)):
                    xt = hip_conv(T['rb.%d.c1.%d' % (i * nk + j, m)], r, _wn(c1), c1.bias, in_slope=0.1)
                    r = hip_conv(T['rb.%d.c2.%d' % (i * nk + j, m)], xt, _wn(c2), c2.bias, resid=r, in_slope=0.1)
            else:
                for m, c in enumerate(rb.convs):
                    r = hip_conv(T['rb.%d.c.%d' % (i * nk + j, m)], r, _wn(c), c.bias, resid=r, in_slope=0.1)
            return r

        # the nk ResBlocks of a stage are independent branches: one stream each (training crops are small launches, see streams.py)
        rs = fan_out([(lambda j=j: branch(j)) for j in range(nk)], x.device, inputs=[x], tag='gen')
        x = _BranchSum.apply(*rs)
    x = hip_conv(T['conv_post'], x, _wn(gen.conv_post), gen.conv_post.bias, in_scale=1.0 / nk, in_slope=0.01)
    return torch.tanh(x)
