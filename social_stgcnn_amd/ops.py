"""Host-side operator layer: torch.autograd.Functions over the C ABI (include/stgcnn_hip.h).

Every function here launches hand-written HIP kernels on the current torch stream and
raises if the library or a GPU tensor is missing -- there is no eager fallback.
"""
import ctypes

import torch

from . import _lib
from ._lib import ModelDesc, check, lib, peds_arg, ptr, require_gpu, stream_ptr


def _adj_layout(adj, n, t, v):
    """adjacency (T,V,V) shared or (N,T,V,V) -> (contiguous-per-scene tensor, batch stride)."""
    if adj.dim() == 3:
        if tuple(adj.shape) != (t, v, v):
            raise ValueError("adjacency %s does not match x (T=%d, V=%d)" % (tuple(adj.shape), t, v))
        return adj.contiguous(), 0
    if adj.dim() == 4:
        if tuple(adj.shape) != (n, t, v, v):
            raise ValueError("adjacency %s does not match x (N=%d, T=%d, V=%d)" % (tuple(adj.shape), n, t, v))
        if adj.stride()[1:] != (v * v, v, 1):
            adj = adj.contiguous()
        return adj, adj.stride(0)
    raise ValueError("adjacency must be (T,V,V) or (N,T,V,V), got %d dims" % adj.dim())


# --------------------------------------------------------------------------------------------
# R1/R2 adjacency build
# --------------------------------------------------------------------------------------------
def adj_build(seq_rel, num_peds=None, normalize=True, out=None):
    """seq_rel (N,V,2,T) fp32 device tensor (any strides) -> nodes (N,T,V,2), adj (N,T,V,V).
    Counterpart of utils.seq_to_graph (utils.py:29-53).  `out` = (nodes, adj) to fill existing buffers (a captured
    step keeps the graph build inside the hipGraph this way)."""
    require_gpu(seq_rel)
    _lib.as_f32(seq_rel, "seq_rel")
    n, v, c, t = seq_rel.shape
    if c != 2:
        raise ValueError("seq_rel must be (N,V,2,T)")
    peds = peds_arg(num_peds, n, seq_rel.device)
    if out is not None:
        nodes, adj = out
        if tuple(nodes.shape) != (n, t, v, 2) or tuple(adj.shape) != (n, t, v, v) or not (
                nodes.is_contiguous() and adj.is_contiguous()):
            raise ValueError("adj_build: out = (nodes (N,T,V,2), adj (N,T,V,V)), contiguous")
    else:
        nodes = torch.empty((n, t, v, 2), device=seq_rel.device, dtype=torch.float32)
        adj = torch.empty((n, t, v, v), device=seq_rel.device, dtype=torch.float32)
    sn, sv, sc, st = seq_rel.stride()
    check(lib().stg_adj_build(ptr(seq_rel), sn, sv, sc, st, ptr(peds), n, v, t, 1 if normalize else 0,
                              ptr(nodes), ptr(adj), stream_ptr()), "stg_adj_build")
    return nodes, adj


# --------------------------------------------------------------------------------------------
# R3 spatial aggregation einsum('nctv,ntvw->nctw')
# --------------------------------------------------------------------------------------------
class _SpatialAgg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, adj, num_peds):
        require_gpu(x, adj)
        _lib.as_f32(x, "x")
        _lib.as_f32(adj, "adj")
        n, c, t, v = x.shape
        adj_c, a_sn = _adj_layout(adj, n, t, v)
        peds = peds_arg(num_peds, n, x.device)
        y = torch.empty((n, c, t, v), device=x.device, dtype=torch.float32)
        sn, sc, st, sv = x.stride()
        check(lib().stg_spatial_agg_fwd(ptr(x), sn, sc, st, sv, ptr(adj_c), a_sn, ptr(peds), n, c, t, v, ptr(y),
                                        stream_ptr()), "stg_spatial_agg_fwd")
        ctx.save_for_backward(adj_c, peds if peds is not None else torch.empty(0))
        ctx.a_sn = a_sn
        ctx.has_peds = peds is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        adj_c, peds = ctx.saved_tensors
        dy = dy.contiguous()
        n, c, t, v = dy.shape
        dx = torch.empty_like(dy)
        check(lib().stg_spatial_agg_bwd(ptr(dy), ptr(adj_c), ctx.a_sn, ptr(peds) if ctx.has_peds else None, n, c, t,
                                        v, ptr(dx), stream_ptr()), "stg_spatial_agg_bwd")
        return dx, None, None


def spatial_agg(x, adj, num_peds=None):
    """y[n,c,t,w] = sum_v x[n,c,t,v] A[n,t,v,w]   (model.py:67); adj (T,V,V) or (N,T,V,V)."""
    return _SpatialAgg.apply(x, adj, num_peds)


# --------------------------------------------------------------------------------------------
# temporal / 1x1 convolution (kt x 1)
# --------------------------------------------------------------------------------------------
class _ConvT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, pad, num_peds):
        require_gpu(x, weight, bias)
        _lib.as_f32(x, "x")
        n, cin, t, v = x.shape
        cout, cin_w, kt, kw = weight.shape
        if cin_w != cin or kw != 1:
            raise ValueError("weight %s does not match input channels %d" % (tuple(weight.shape), cin))
        to = t + 2 * pad - kt + 1
        peds = peds_arg(num_peds, n, x.device)
        w = weight.contiguous()
        b = bias.contiguous() if bias is not None else None
        y = torch.empty((n, cout, to, v), device=x.device, dtype=torch.float32)
        sn, sc, st, sv = x.stride()
        check(lib().stg_conv_t_fwd(ptr(x), sn, sc, st, sv, ptr(w), ptr(b), ptr(peds), n, cin, cout, t, v, kt, pad,
                                   ptr(y), stream_ptr()), "stg_conv_t_fwd")
        ctx.save_for_backward(x, w, peds if peds is not None else torch.empty(0))
        ctx.has_peds = peds is not None
        ctx.has_bias = bias is not None
        ctx.pad = pad
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, peds = ctx.saved_tensors
        dy = dy.contiguous()
        n, cin, t, v = x.shape
        cout, _, kt, _ = w.shape
        dx = torch.empty((n, cin, t, v), device=x.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        dw = torch.zeros_like(w)
        db = torch.zeros(cout, device=x.device, dtype=torch.float32) if ctx.has_bias else None
        sn, sc, st, sv = x.stride()
        check(lib().stg_conv_t_bwd(ptr(x), sn, sc, st, sv, ptr(w), ptr(dy), ptr(peds) if ctx.has_peds else None, n,
                                   cin, cout, t, v, kt, ctx.pad, ptr(dx), ptr(dw), ptr(db), stream_ptr()),
              "stg_conv_t_bwd")
        return dx, dw, db, None, None


def conv_t(x, weight, bias, pad=0, num_peds=None):
    """nn.Conv2d(Cin, Cout, (kt,1), padding=(pad,0)) forward/backward (model.py:55-62)."""
    return _ConvT.apply(x, weight, bias, pad, num_peds)


# --------------------------------------------------------------------------------------------
# R4/R5 fused st_gcn / social_stgcnn
# --------------------------------------------------------------------------------------------
class FlatPack:
    """Keeps a list of tensors as views of ONE flat fp32 device buffer (the layout the fused
    kernels read).  `ensure()` is cheap when nothing moved; it re-packs after .to(), a
    load_state_dict that replaced storage, or a child module packing itself."""

    def __init__(self):
        self.flat = None

    def ensure(self, tensors):
        total = sum(t.numel() for t in tensors)
        dev = tensors[0].device
        flat = self.flat
        ok = flat is not None and flat.device == dev and flat.numel() == total
        if ok:
            base = flat.data_ptr()
            off = 0
            for t in tensors:
                if t.data_ptr() != base + 4 * off or not t.is_contiguous():
                    ok = False
                    break
                off += t.numel()
        if not ok:
            flat = torch.empty(total, device=dev, dtype=torch.float32)
            off = 0
            with torch.no_grad():
                for t in tensors:
                    n = t.numel()
                    flat[off:off + n].copy_(t.detach().reshape(-1))
                    t.data = flat[off:off + n].view(t.shape)
                    off += n
            self.flat = flat
        return self.flat


# Launch options of the fused model entry points (stg_model_desc.flags / .wg_waves).  Host-side switches: the
# library itself reads no environment and keeps no state, every choice travels in the descriptor.
#   wg_path     run the workgroup-per-scene kernels even where the wave-per-scene path fits (tests cover both)
#   split_bf16  TXP input-gradient GEMMs on bf16 MFMAs with hi/lo-split operands (opt-in, fp32 in / out)
#   wg_waves    0 = auto, or 1 / 2 / 4 / 8 waves per scene in the workgroup-per-scene kernels
#   wave_path   keep the wave-per-scene kernels where the library would pick the workgroup-per-scene ones for a small batch
#               (only reachable with f32_mfma / split_bf16: the exact-bf16 kernels serve small batches with finer teams)
#   bf16_store  bf16 storage of the saved TXP activations and of the dz hand-off (STG_OPT_BF16_STORE): fp32 forward
#               result, ~1e-3 relative error in the TXP weight / slope gradients; wave-per-scene path only
#   f32_mfma    the fp32-MFMA kernels where the default runs the exact bf16-pipe ones (STG_OPT_F32_MFMA): A/B measurements
class KernelOptions(dict):
    """The launch options of ONE model: `model.options = ops.KernelOptions(bf16_store=True)` makes that model (and the
    Trainer / EpochRunner driving it) run with its own choices, whatever other models in the process do.  A model without
    an `options` attribute of its own uses the process-wide defaults, `ops.OPTIONS`."""
    DEFAULTS = {"wg_path": False, "split_bf16": False, "wg_waves": 0, "wave_path": False, "bf16_store": False,
                "f32_mfma": False}

    def __init__(self, **kw):
        bad = set(kw) - set(self.DEFAULTS)
        if bad:
            raise KeyError("unknown kernel options %s" % sorted(bad))
        super().__init__(self.DEFAULTS)
        self.update(kw)


OPTIONS = KernelOptions()           # process-wide defaults (models without their own `options`)


def make_desc(n_stgcnn, n_txpcnn, c_in, c_out, t_obs, t_pred, kt, residual0, use_mdn, training,
              eps=1e-5, momentum=0.1, options=None):
    o = OPTIONS if options is None else options
    flags = ((_lib.OPT_WG_PATH if o["wg_path"] else 0) | (_lib.OPT_SPLIT_BF16 if o["split_bf16"] else 0)
             | (_lib.OPT_WAVE_PATH if o["wave_path"] else 0)
             | (_lib.OPT_BF16_STORE if o["bf16_store"] else 0)
             | (_lib.OPT_F32_MFMA if o["f32_mfma"] else 0))
    return ModelDesc(n_stgcnn, n_txpcnn, c_in, c_out, t_obs, t_pred, kt, residual0, 1 if use_mdn else 0,
                     1 if training else 0, eps, momentum, flags, int(o["wg_waves"]))


class KernelTimer:
    """Per-kernel device time of the fused forward / backward entry points: HIP events recorded by the library
    itself on the launch stream between its kernels (`events` argument of stg_model_fwd / stg_model_bwd).
    Attached to ONE model (`model.timer = ops.KernelTimer()`, bench.py's roofline leg); no timer, no events."""
    MAX_KERNELS = 8

    def __init__(self):
        self.calls = {"model_fwd": [], "model_bwd": []}

    def events(self, name):
        ev = _lib.HipEvents(self.MAX_KERNELS + 1)
        self.calls[name].append(ev)
        return ev

    def kernel_ms(self, name):
        """mean duration (ms) of the k-th kernel of entry point `name` over the recorded calls; events the entry
        point did not reach are dropped (their interval cannot be read)."""
        rows = []
        for ev in self.calls[name]:
            row = []
            hip = ev.hip()
            hip.hipEventSynchronize(ev.arr[0])
            for k in range(1, ev.n):
                ms = ctypes.c_float()
                if hip.hipEventElapsedTime(ctypes.byref(ms), ev.arr[k - 1], ev.arr[k]) != 0:
                    hip.hipGetLastError()          # (an unrecorded handle: clear the runtime's sticky error)
                    break
                row.append(ms.value)
            rows.append(row)
        if not rows:
            return []
        n = min(len(r) for r in rows)
        return [sum(r[k] for r in rows) / len(rows) for k in range(n)]

    def mean_ms(self, name):
        return sum(self.kernel_ms(name))


def _timer_of(holder):
    return getattr(holder, "timer", None) if holder is not None else None


class _FusedModel(torch.autograd.Function):
    """x (N,Cin,T,V), adj -> y.  Extra (non-differentiable) arguments carry the packed buffers."""

    @staticmethod
    def forward(ctx, x, adj, num_peds, desc, flat_params, flat_buffers, nbt, dead, holder, grad_on, *params):
        require_gpu(x, adj, flat_params, flat_buffers)
        _lib.as_f32(x, "x")
        _lib.as_f32(adj, "adj")
        L = lib()
        n, cin, t, v = x.shape
        if cin != desc.c_in or t != desc.t_obs:
            raise ValueError("input (N,%d,%d,V) does not match the model (input_feat=%d, seq_len=%d)"
                             % (cin, t, desc.c_in, desc.t_obs))
        adj_c, a_sn = _adj_layout(adj, n, t, v)
        peds = peds_arg(num_peds, n, x.device)
        training = desc.bn_mode == 1
        # needs_input_grad reflects requires_grad, not the grad mode (and forward() itself always runs with grad
        # disabled): the caller's grad mode comes in as `grad_on`, so that under no_grad nothing is saved
        need_grad = grad_on and any(ctx.needs_input_grad)
        if need_grad and ctx.needs_input_grad[0] and not (desc.flags & _lib.OPT_WG_PATH):
            # an input gradient is only computed by the workgroup-per-scene kernels (both passes must agree)
            fields = {f: getattr(desc, f) for f, _ in ModelDesc._fields_}
            fields["flags"] |= _lib.OPT_WG_PATH
            desc = ModelDesc(**fields)
        out_t = desc.t_pred if desc.n_txpcnn > 0 else desc.t_obs
        y = torch.empty((n, desc.c_out, out_t, v), device=x.device, dtype=torch.float32)
        ws_floats = 0                  # (diagnostics: activation workspace this forward allocated; 0 = inference)
        ws = None
        if need_grad:
            wsf = L.stg_model_ws_floats(ctypes.byref(desc), v)
            if wsf < 0:
                check(int(wsf), "stg_model_ws_floats")
            tail = L.stg_model_ws_tail_floats(ctypes.byref(desc), n, v)
            if tail < 0:
                check(int(tail), "stg_model_ws_tail_floats")
            ws = torch.empty(n * wsf + tail, device=x.device, dtype=torch.float32)
            ws_floats = ws.numel()
        stats = None
        if training:
            sf = L.stg_model_stat_floats(ctypes.byref(desc))
            stats = torch.empty((n, max(int(sf), 1)), device=x.device, dtype=torch.float32)
        scr = None
        nscr = L.stg_model_fwd_scratch_floats(ctypes.byref(desc), n, v)
        if nscr < 0:
            check(int(nscr), "stg_model_fwd_scratch_floats")
        if nscr > 0:
            scr = torch.empty(int(nscr), device=x.device, dtype=torch.float32)
        if holder is not None:
            holder.last_ws_floats = ws_floats
            holder._last_fwd_scratch = scr  # diagnostics only (STG_STAMPS=1 reads the stamp tail)
        sn, sc, st, sv = x.stride()
        timer = _timer_of(holder)
        ev = timer.events("model_fwd") if timer is not None else None
        check(L.stg_model_fwd(ctypes.byref(desc), ptr(flat_params), ptr(flat_buffers), ptr(x), sn, sc, st, sv,
                              ptr(adj_c), a_sn, ptr(peds), n, v, ptr(y), ptr(ws), ptr(stats), ptr(scr),
                              ev.arr if ev else None, ev.n if ev else 0, stream_ptr()),
              "stg_model_fwd")
        if training and holder is not None and getattr(holder, "_defer_bn_fold", False):
            # the trainer folds the running statistics in the step's tail launch (ops.train_tail)
            holder._pending_bn = (desc, stats, peds, n, flat_buffers, nbt)
        elif training:
            arr = (ctypes.c_void_p * len(nbt))(*[b.data_ptr() for b in nbt])
            # nbt[k] counts forwards of BatchNorm k; buffers are interleaved (mean, var) per BatchNorm, the
            # kernel bumps counter i for statistic row i < len(nbt): pass one pointer per BatchNorm.
            check(L.stg_bn_fold(ctypes.byref(desc), ptr(stats), ptr(peds), n, ptr(flat_buffers), arr, len(nbt),
                                stream_ptr()), "stg_bn_fold")
        if holder is not None:
            # what a backward needs, for the trainer's fused loss + backward (backward_from_target)
            holder._fwd_state = (desc, a_sn, dead, [tuple(p.shape) for p in params], flat_params, flat_buffers,
                                 (x, adj_c, peds, ws)) if need_grad else None
        ctx.desc = desc
        ctx.a_sn = a_sn
        ctx.dead = dead
        ctx.holder = holder
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.flat_params = flat_params
        ctx.flat_buffers = flat_buffers
        ctx.tensors = (x, adj_c, peds, ws)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = lib()
        desc = ctx.desc
        x, adj_c, peds, ws = ctx.tensors
        if ws is None:
            raise RuntimeError("backward through a forward that saved no activations")
        dy = dy.contiguous()
        n, cin, t, v = x.shape
        np_ = int(L.stg_model_param_count(ctypes.byref(desc)))
        n_scratch = L.stg_model_bwd_scratch_floats(ctypes.byref(desc), n, v)
        if n_scratch < 0:
            check(int(n_scratch), "stg_model_bwd_scratch_floats")
        slabs = torch.empty(int(n_scratch), device=x.device, dtype=torch.float32)
        grad = torch.empty(np_, device=x.device, dtype=torch.float32)
        dx = torch.empty((n, cin, t, v), device=x.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        sn, sc, st, sv = x.stride()
        timer = _timer_of(ctx.holder)
        ev = timer.events("model_bwd") if timer is not None else None
        check(L.stg_model_bwd(ctypes.byref(desc), ptr(ctx.flat_params), ptr(ctx.flat_buffers), ptr(x), sn, sc, st,
                              sv, ptr(adj_c), ctx.a_sn, ptr(peds), n, v, ptr(dy), ptr(ws), ptr(slabs), ptr(grad),
                              ptr(dx), ev.arr if ev else None, ev.n if ev else 0, stream_ptr()), "stg_model_bwd")
        grads = []
        off = 0
        for i, shp in enumerate(ctx.shapes):
            cnt = 1
            for s in shp:
                cnt *= s
            grads.append(None if i in ctx.dead else grad[off:off + cnt].view(shp))
            off += cnt
        if ctx.holder is not None:
            ctx.holder._flat_grad = grad       # the trainer all-reduces / applies this buffer directly
            ctx.holder._fwd_state = None
        return (dx, None, None, None, None, None, None, None, None, None, *grads)


def backward_from_target(holder, y, target, weights=None, step=None):
    """Loss + backward of the fused forward `holder` (a model) just ran, in the backward's own launches
    (stg_model_bwd_nll): per-scene bivariate losses (N,) are returned, the parameter gradients of
    sum_n weights[n] * loss_n land in `holder._flat_grad` and in every live parameter's .grad (views of it).
    Returns None -- nothing launched -- where the library has no fused form (STG_OPT_SPLIT_BF16, a model without a
    TXP-CNN): the caller then takes the separate loss kernel + autograd backward.

    step = (pending_bn, lr, lr_dev): the rest of a single-rank training step rides in the same launches
    (stg_model_bwd_step: SGD without clipping on the flat parameters, the BatchNorm fold of `pending_bn` -- what a
    forward with a deferred fold left behind -- and the reported loss); returns (losses, total) then."""
    st = getattr(holder, "_fwd_state", None)
    if st is None:
        raise RuntimeError("backward_from_target: no fused forward with saved activations to start from")
    desc, a_sn, dead, shapes, flat_params, flat_buffers, (x, adj_c, peds, ws) = st
    L = lib()
    n, cin, t, v = x.shape
    if tuple(y.shape) != (n, 5, desc.t_pred, v) or not y.is_contiguous():
        return None
    target = target.to(torch.float32).contiguous()
    if tuple(target.shape) != (n, desc.t_pred, v, 2):
        raise ValueError("backward_from_target: target (N,P,V,2) expected, got %s" % (tuple(target.shape),))
    w = weights.to(torch.float32).contiguous() if weights is not None else None
    np_ = int(L.stg_model_param_count(ctypes.byref(desc)))
    n_scratch = L.stg_model_bwd_scratch_floats(ctypes.byref(desc), n, v)
    if n_scratch < 0:
        check(int(n_scratch), "stg_model_bwd_scratch_floats")
    slabs = torch.empty(int(n_scratch), device=x.device, dtype=torch.float32)
    grad = torch.empty(np_, device=x.device, dtype=torch.float32)
    losses = torch.empty(n, device=x.device, dtype=torch.float32)
    sn, sc, st_, sv = x.stride()
    timer = _timer_of(holder)
    ev = timer.events("model_bwd") if timer is not None else None
    total = None
    if step is None:
        rc = L.stg_model_bwd_nll(ctypes.byref(desc), ptr(flat_params), ptr(flat_buffers), ptr(x), sn, sc, st_, sv,
                                 ptr(adj_c), a_sn, ptr(peds), n, v, ptr(y), ptr(target), ptr(w), ptr(losses), ptr(ws),
                                 ptr(slabs), ptr(grad), ev.arr if ev else None, ev.n if ev else 0, stream_ptr())
    else:
        pending_bn, lr, lr_dev = step
        total = torch.empty(1, device=x.device, dtype=torch.float32)
        tail = _lib.StepTail()
        tail.params, tail.lr_dev, tail.lr = ptr(flat_params), ptr(lr_dev), float(lr)
        tail.total = ptr(total)
        if pending_bn is not None:
            _, stats, _, _, bn_buffers, nbt = pending_bn
            arr = (ctypes.c_void_p * len(nbt))(*[b.data_ptr() for b in nbt])
            tail.stats, tail.buffers, tail.nbt, tail.n_bn = ptr(stats), ptr(bn_buffers), arr, len(nbt)
        rc = L.stg_model_bwd_step(ctypes.byref(desc), ptr(flat_params), ptr(flat_buffers), ptr(x), sn, sc, st_, sv,
                                  ptr(adj_c), a_sn, ptr(peds), n, v, ptr(y), ptr(target), ptr(w), ptr(losses), ptr(ws),
                                  ptr(slabs), ptr(grad), ctypes.addressof(tail), ev.arr if ev else None,
                                  ev.n if ev else 0, stream_ptr())
    if rc == _lib.EUNSUPPORTED:
        if ev is not None:
            timer.calls["model_bwd"].pop()
        return None
    check(rc, "stg_model_bwd_nll" if step is None else "stg_model_bwd_step")
    off = 0
    for i, (p, shp) in enumerate(zip(holder._tensors()[0], shapes)):
        cnt = p.numel()
        p.grad = None if i in dead else grad[off:off + cnt].view(shp)
        off += cnt
    holder._flat_grad = grad
    holder._fwd_state = None
    return losses if step is None else (losses, total[0])


def fused_model(x, adj, num_peds, desc, flat_params, flat_buffers, nbt, dead, params, holder=None):
    return _FusedModel.apply(x, adj, num_peds, desc, flat_params, flat_buffers, nbt, dead, holder,
                             torch.is_grad_enabled(), *params)


# --------------------------------------------------------------------------------------------
# R6 bivariate Gaussian NLL
# --------------------------------------------------------------------------------------------
class _BivariateNLL(torch.autograd.Function):
    """pred (N,P,V,5) (any strides), target (N,P,V,2) -> loss (N,)."""

    @staticmethod
    def forward(ctx, pred, target, num_peds):
        require_gpu(pred, target)
        _lib.as_f32(pred, "V_pred")
        n, p, v, f = pred.shape
        if f != 5 or tuple(target.shape) != (n, p, v, 2):
            raise ValueError("bivariate_loss: V_pred (..,P,V,5) / V_trgt (..,P,V,2) expected, got %s / %s"
                             % (tuple(pred.shape), tuple(target.shape)))
        target = target.to(torch.float32).contiguous()
        peds = peds_arg(num_peds, n, pred.device)
        loss = torch.empty(n, device=pred.device, dtype=torch.float32)
        need = ctx.needs_input_grad[0]
        grad = torch.empty((n, 5, p, v), device=pred.device, dtype=torch.float32) if need else None
        sn, sp, sv, sf = pred.stride()
        check(lib().stg_nll_fwd(ptr(pred), sn, sf, sp, sv, ptr(target), ptr(peds), None, n, p, v, ptr(loss), ptr(grad),
                                stream_ptr()), "stg_nll_fwd")
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, gloss):
        grad = ctx.grad
        n, _, p, v = grad.shape
        gloss = gloss.to(torch.float32).contiguous()
        out = torch.empty_like(grad)
        check(lib().stg_nll_bwd(ptr(grad), ptr(gloss), n, p, v, ptr(out), stream_ptr()), "stg_nll_bwd")
        # (N,5,P,V) buffer viewed in the caller's (N,P,V,5) index order
        return out.permute(0, 2, 3, 1), None, None


def bivariate_nll(pred, target, num_peds=None):
    return _BivariateNLL.apply(pred, target, num_peds)


def bivariate_nll_with_grad(y, target, num_peds=None, weights=None):
    """Trainer fast path: y (N,5,P,V) model output (contiguous), target (N,P,V,2) ->
    (per-scene losses (N,), d(sum_n w_n loss_n)/dy (N,5,P,V)) from ONE nll_fwd launch."""
    require_gpu(y, target)
    n, f, p, v = y.shape
    if f != 5 or tuple(target.shape) != (n, p, v, 2):
        raise ValueError("bivariate_nll_with_grad: y (N,5,P,V) / target (N,P,V,2) expected")
    target = target.to(torch.float32).contiguous()
    peds = peds_arg(num_peds, n, y.device)
    loss = torch.empty(n, device=y.device, dtype=torch.float32)
    grad = torch.empty((n, 5, p, v), device=y.device, dtype=torch.float32)
    w = weights.to(torch.float32).contiguous() if weights is not None else None
    sn, sf, sp, sv = y.stride()
    check(lib().stg_nll_fwd(ptr(y), sn, sf, sp, sv, ptr(target), ptr(peds), ptr(w), n, p, v, ptr(loss), ptr(grad),
                            stream_ptr()), "stg_nll_fwd")
    return loss, grad


def sgd_step(flat_params, flat_grads, lr):
    """p -= lr * g on the flat buffers (train.py:197 SGD without momentum)."""
    require_gpu(flat_params, flat_grads)
    check(lib().stg_sgd_step(ptr(flat_params), ptr(flat_grads), flat_params.numel(), float(lr), stream_ptr()),
          "stg_sgd_step")


def optim_step(flat_params, flat_grads, lr, max_norm=None, lr_dev=None, grad_norm=None):
    """clip_grad_norm_(max_norm) + SGD(lr) on the flat buffers in one launch (train.py:71-74,197).  `lr_dev`
    (1-element device tensor) overrides `lr` -- the form a captured hipGraph needs to follow StepLR; `grad_norm`
    (1-element device tensor) receives the unclipped gradient norm.  The gradients are scaled in place."""
    require_gpu(flat_params, flat_grads)
    check(lib().stg_optim_step(ptr(flat_params), ptr(flat_grads), flat_params.numel(), ptr(lr_dev), float(lr),
                               float(max_norm) if max_norm is not None else 0.0, ptr(grad_norm), stream_ptr()),
          "stg_optim_step")


def train_tail(pending_bn, losses, weights, flat_params, flat_grads, lr, max_norm=None, lr_dev=None):
    """The tail of a single-rank step in one launch (stg_train_tail): BatchNorm fold of `pending_bn` (what a fused
    forward with a deferred fold left behind), the reported loss sum_n w_n loss_n, clip + SGD.  Returns the loss as a
    0-d device tensor."""
    require_gpu(losses, flat_params, flat_grads)
    desc, stats, peds, n, flat_buffers, nbt = pending_bn
    arr = (ctypes.c_void_p * len(nbt))(*[b.data_ptr() for b in nbt])
    w = weights.to(torch.float32).contiguous() if weights is not None else None
    out = torch.empty(1, device=losses.device, dtype=torch.float32)
    check(lib().stg_train_tail(ctypes.byref(desc), ptr(stats), ptr(peds), n, ptr(flat_buffers), arr, len(nbt),
                               ptr(losses), ptr(w), ptr(out), ptr(flat_params), ptr(flat_grads), flat_params.numel(),
                               ptr(lr_dev), float(lr), float(max_norm) if max_norm is not None else 0.0, None,
                               stream_ptr()), "stg_train_tail")
    return out[0]


def dp_pack(flat_grad, bn_before, bn_after, num_peds, n_scenes, momentum, rank, world, pack):
    """[gradient | this rank's BatchNorm contribution and scene count] into the ONE buffer a data-parallel step
    all-reduces (stg_dp_pack).  pack: n_params + world * (n_buffers + 1) floats."""
    require_gpu(flat_grad, bn_before, bn_after, pack)
    peds = peds_arg(num_peds, n_scenes, flat_grad.device)
    check(lib().stg_dp_pack(ptr(flat_grad), ptr(bn_before), ptr(bn_after), ptr(peds), int(n_scenes), float(momentum),
                            int(rank), int(world), flat_grad.numel(), bn_before.numel(), ptr(pack), stream_ptr()),
          "stg_dp_pack")


def dp_fold(pack, bn_before, momentum, rank, world, n_params, buffers, nbt=()):
    """the exact sequential fold of the running statistics over the ranks from the all-reduced pack (stg_dp_fold);
    `nbt` (the num_batches_tracked tensors) also receive the other ranks' scene counts from the pack."""
    require_gpu(pack, bn_before, buffers)
    arr = (ctypes.c_void_p * len(nbt))(*[b.data_ptr() for b in nbt]) if len(nbt) else None
    check(lib().stg_dp_fold(ptr(pack), ptr(bn_before), float(momentum), int(rank), int(world), int(n_params),
                            bn_before.numel(), ptr(buffers), arr, len(nbt), stream_ptr()), "stg_dp_fold")


def weighted_sum(values, weights=None):
    """sum_n w_n v_n as a 1-element device tensor (one launch, fixed summation order)."""
    require_gpu(values)
    values = values.contiguous()
    w = weights.to(torch.float32).contiguous() if weights is not None else None
    out = torch.empty(1, device=values.device, dtype=torch.float32)
    check(lib().stg_weighted_sum(ptr(values), ptr(w), values.numel(), ptr(out), stream_ptr()), "stg_weighted_sum")
    return out[0]


def best_of_k(y, target_rel, obs_last=None, num_peds=None, k=20, noise=None, seed=0):
    """Evaluation tail of test.py:59-123 on the device: y (N,5,P,V) model output (any strides), target_rel
    (N,P,V,2), obs_last (N,V,2) or None, noise (K,N,P,V,2) standard normals or None (in-kernel Philox stream keyed
    by `seed`).  Returns per-pedestrian (min ADE, min FDE), each (N,V) with zeros in padded slots."""
    require_gpu(y, target_rel)
    n, f, p, v = y.shape
    if f != 5 or tuple(target_rel.shape) != (n, p, v, 2):
        raise ValueError("best_of_k: y (N,5,P,V) / target_rel (N,P,V,2) expected")
    y = y.to(torch.float32)
    target_rel = target_rel.to(torch.float32).contiguous()
    if obs_last is not None:
        if tuple(obs_last.shape) != (n, v, 2):
            raise ValueError("best_of_k: obs_last (N,V,2) expected")
        obs_last = obs_last.to(device=y.device, dtype=torch.float32).contiguous()
    if noise is not None:
        if tuple(noise.shape) != (k, n, p, v, 2):
            raise ValueError("best_of_k: noise (K,N,P,V,2) expected")
        noise = noise.to(device=y.device, dtype=torch.float32).contiguous()
    peds = peds_arg(num_peds, n, y.device)
    ade = torch.empty((n, v), device=y.device, dtype=torch.float32)
    fde = torch.empty((n, v), device=y.device, dtype=torch.float32)
    sn, sf, sp, sv = y.stride()
    check(lib().stg_bestofk_eval(ptr(y), sn, sf, sp, sv, ptr(target_rel), ptr(obs_last), ptr(peds), ptr(noise),
                                 int(seed) & 0xFFFFFFFFFFFFFFFF, n, p, v, int(k), ptr(ade), ptr(fde), stream_ptr()),
          "stg_bestofk_eval")
    return ade, fde


def scene_order(num_peds, v):
    """Scene indices sorted by pedestrian count (clamped to [0, v]) descending, stable: the schedule the fused
    kernels use for ragged batches (`stg_scene_order`).  num_peds: int32 device tensor (N,).
    Returns (order (N,), key_start (v+2,)): scenes with at most x pedestrians are order[key_start[v-x]:]."""
    require_gpu(num_peds)
    n = num_peds.numel()
    peds = peds_arg(num_peds, n, num_peds.device)
    order = torch.empty(n, device=peds.device, dtype=torch.int32)
    key_start = torch.empty(int(v) + 2, device=peds.device, dtype=torch.int32)
    if n < 2:
        order.zero_()
        key_start.fill_(n)
        if n == 1:
            key_start[: int(v) - max(0, min(int(v), int(peds[0]))) + 1] = 0
        return order, key_start
    check(lib().stg_scene_order(ptr(peds), n, int(v), ptr(order), ptr(key_start), stream_ptr()), "stg_scene_order")
    return order, key_start
