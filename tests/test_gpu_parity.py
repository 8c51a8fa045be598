"""GPU parity tests (-m gpu): every HIP entry point, called through the C ABI (ctypes),
against the CPU oracle (oracle/stgcnn_oracle.py) and the reference-generated fixtures in
tests/golden/.  Tolerances: fp32; the north-star bar for the bivariate parameters is 1e-4
absolute, the tests hold the kernels to 2e-5 or tighter and state each bound where it is used."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

VS = (2, 3, 5, 8, 17, 32, 57)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda", 0)


def _oracle():
    from oracle import stgcnn_oracle as O
    return O


def _state(npz, prefix=""):
    return {k[len(prefix):]: torch.from_numpy(np.array(npz[k])) for k in npz.files if k.startswith(prefix)}


def _model(dev, state=None, seed=None, **kw):
    from social_stgcnn_amd.model import social_stgcnn
    cfg = dict(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12)
    cfg.update(kw)
    if seed is not None:
        torch.manual_seed(seed)
    m = social_stgcnn(**cfg)
    if state is not None:
        m.load_state_dict(state)
    return m.to(dev)


def _maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


# Conv biases that feed a train-mode BatchNorm (gcn.conv.bias, tcn.2.bias, residual.0.bias) have an
# exactly-zero true gradient: the normalisation removes any per-channel shift.  Both the reference and the
# kernels hold fp32 rounding noise of the same sum there (order eps * |weight gradient|), so they are
# compared on the scale of the sibling weight gradient instead of their own ~1e-8 magnitude.
_ZERO_GRAD_BIASES = ("gcn.conv.bias", "tcn.2.bias", "residual.0.bias")


def _grad_errors(named_got, ref_of):
    """named_got: iterable of (name, grad tensor or None); ref_of(name) -> numpy array or None (dead).
    Returns {name: relative error} for live parameters; asserts dead ones are None on both sides."""
    named_got = list(named_got)
    refs = {name: ref_of(name) for name, _ in named_got}
    out = {}
    for name, got in named_got:
        ref = refs[name]
        if ref is None:
            assert got is None, name
            continue
        assert got is not None, name
        scale = max(1e-3, float(np.abs(ref).max()))
        if name.endswith(_ZERO_GRAD_BIASES):
            sib = refs.get(name[:-4] + "weight")
            if sib is not None:
                scale = max(scale, float(np.abs(sib).max()))
        err = _maxdiff(got.cpu().numpy(), ref) / scale
        # pure-noise entries get a 10x looser bar (they scale with depth / batch, not with the kernels)
        out[name] = err / 10.0 if name.endswith(_ZERO_GRAD_BIASES) else err
    return out


# ------------------------------------------------------------------------------------------
def test_library_loaded_and_mfma_operand_maps(dev):
    from social_stgcnn_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(0)
    for K in (4, 72, 108):
        a = rng.standard_normal((16, K)).astype(np.float32)
        b = rng.standard_normal((K, 16)).astype(np.float32)      # asymmetric on purpose
        ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
        tc = torch.zeros(16, 16, device=dev)
        _lib.check(L.stg_selftest_mfma(_lib.ptr(ta), _lib.ptr(tb), K, _lib.ptr(tc), _lib.stream_ptr()), "selftest")
        ref = a.astype(np.float64) @ b.astype(np.float64)
        assert _maxdiff(tc.cpu().numpy(), ref) < 1e-4 * max(1.0, np.abs(ref).max())


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("v", VS + ("tie",))
def test_adj_build_golden(dev, v):
    """R1/R2 against the reference's seq_to_graph output (fixture) -- bound 1e-6 absolute."""
    from social_stgcnn_amd.utils import seq_to_graph
    g = load_golden("adj_cases.npz")
    rel = torch.from_numpy(g["rel_%s" % v])
    nodes, adj = seq_to_graph(None, rel, True)            # CPU in -> CPU out, computed on the GPU
    assert nodes.device.type == "cpu"
    assert np.array_equal(nodes.numpy(), g["nodes_%s" % v])
    assert _maxdiff(adj.numpy(), g["lap_%s" % v]) < 1e-6
    assert np.all(adj.numpy()[0] == 0)


def test_adj_build_batched_ragged(dev):
    """padded batch with num_peds: padded rows/cols are zero, valid block == per-scene result;
    raw (un-normalised) adjacency has unit diagonal and the exact ==0 rule."""
    from social_stgcnn_amd import ops
    O = _oracle()
    g = load_golden("adj_cases.npz")
    vs = [2, 17, 5, 57, 32, 3, 8]
    vmax = 60
    rel = torch.zeros(len(vs), vmax, 2, 8)
    for i, v in enumerate(vs):
        rel[i, :v] = torch.from_numpy(g["rel_%d" % v])
    rel[:, 58:] = 7.0        # garbage in padded slots must be ignored
    nodes, adj = ops.adj_build(rel.to(dev), num_peds=vs, normalize=True)
    nodes, adj = nodes.cpu().numpy(), adj.cpu().numpy()
    for i, v in enumerate(vs):
        assert _maxdiff(adj[i, :, :v, :v], g["lap_%d" % v]) < 1e-6
        assert np.all(adj[i, :, v:, :] == 0) and np.all(adj[i, :, :, v:] == 0)
        assert np.array_equal(nodes[i, :, :v], g["nodes_%d" % v]) and np.all(nodes[i, :, v:] == 0)
    _, raw = ops.adj_build(torch.from_numpy(g["rel_tie"]).unsqueeze(0).to(dev), normalize=False)
    raw = raw[0].cpu().numpy()
    assert np.all(raw[:, np.arange(4), np.arange(4)] == 1)
    assert raw[2, 0, 1] == 0 and raw[2, 1, 0] == 0          # equal velocities -> no edge
    # odd V exercises the scalar-store path
    rel5 = torch.from_numpy(g["rel_5"]).unsqueeze(0)
    _, a5 = ops.adj_build(rel5.to(dev))
    assert _maxdiff(a5[0].cpu().numpy(), O.seq_to_graph_np(g["rel_5"])[1]) < 1e-6
    # batch padded to exactly 32: the row-per-thread kernel -- bitwise equal to the generic kernel (same batch padded
    # to 36), normalised and raw, garbage in the padded slots ignored
    vs32 = [2, 17, 5, 32, 3, 8, 0, 31]
    rel32 = torch.zeros(len(vs32), 36, 2, 8)
    for i, v in enumerate(vs32):
        if v in (2, 17, 5, 32, 3, 8):
            rel32[i, :v] = torch.from_numpy(g["rel_%d" % v])
        elif v:
            rel32[i, :v] = torch.from_numpy(g["rel_32"])[:v]
    rel32[:, 32:] = 7.0
    for norm in (True, False):
        n_a, a_a = ops.adj_build(rel32[:, :32].contiguous().to(dev), num_peds=vs32, normalize=norm)
        n_b, a_b = ops.adj_build(rel32.to(dev), num_peds=vs32, normalize=norm)
        assert torch.equal(a_a, a_b[:, :, :32, :32]) and torch.equal(n_a, n_b[:, :, :32])
    assert _maxdiff(a_a[3].cpu().numpy() * 0 + ops.adj_build(rel32[3:4, :32].contiguous().to(dev))[1][0].cpu().numpy(),
                    g["lap_32"]) < 1e-6
    # more than 2048 scenes of 32 slots leave through a half-size LDS tile (two phases): bitwise equal to the one-phase kernel
    big32 = rel32[:, :32].repeat(2056 // len(vs32) + 1, 1, 1, 1)[:2056].contiguous().to(dev)
    vbig32 = (vs32 * (2056 // len(vs32) + 1))[:2056]
    for norm in (True, False):
        n_s, a_s = ops.adj_build(rel32[:, :32].contiguous().to(dev), num_peds=vs32, normalize=norm)
        n_l, a_l = ops.adj_build(big32, num_peds=vbig32, normalize=norm)
        assert torch.equal(a_s, a_l[:len(vs32)]) and torch.equal(n_s, n_l[:len(vs32)])
        assert torch.equal(a_l[:len(vs32)], a_l[2048:2048 + len(vs32)])
    # small batches (fewer than 1024 scenes) run a workgroup per (scene, time step), large ones one per scene: bitwise equal
    # (the ragged batch above tiled to 1024 scenes), normalised and raw, 16-byte and scalar stores (V = 60 / 59)
    for vpad in (60, 59):
        relp = torch.zeros(len(vs), vpad, 2, 8)
        relp[:, :min(vpad, vmax)] = rel[:, :min(vpad, vmax)]
        big = relp.repeat(1024 // len(vs) + 1, 1, 1, 1)[:1024].contiguous().to(dev)
        vbig = (vs * (1024 // len(vs) + 1))[:1024]
        for norm in (True, False):
            n_s, a_s = ops.adj_build(relp.to(dev), num_peds=vs, normalize=norm)
            n_l, a_l = ops.adj_build(big, num_peds=vbig, normalize=norm)
            assert torch.equal(a_s, a_l[:len(vs)]) and torch.equal(n_s, n_l[:len(vs)])
            assert torch.equal(a_l[:len(vs)], a_l[1022 - 1022 % len(vs) - len(vs):1022 - 1022 % len(vs)])


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,c,v,shared", [(3, 5, 32, False), (2, 2, 7, False), (4, 5, 57, True), (1, 11, 12, False),
                                           (2, 5, 64, False), (3, 3, 16, False), (5, 2, 8, True), (2, 9, 32, True)])
def test_spatial_agg(dev, n, c, v, shared):
    """R3 einsum('nctv,ntvw->nctw') forward and dx; fp32 reference = torch.einsum on the CPU, 2e-5."""
    from social_stgcnn_amd import ops
    g = torch.Generator().manual_seed(v)
    x = torch.randn(n, v, c, 8, generator=g).permute(0, 2, 3, 1)       # non-contiguous like train.py:48
    A = torch.randn(8, v, v, generator=g) if shared else torch.randn(n, 8, v, v, generator=g)
    peds = None if shared else [max(1, v - 3 * i) for i in range(n)]
    xr = x.clone().requires_grad_(True)
    eq = 'nctv,tvw->nctw' if shared else 'nctv,ntvw->nctw'
    if peds is not None:
        # reference on the valid block of every scene only
        ref = torch.zeros(n, c, 8, v)
        parts = []
        for i, p in enumerate(peds):
            parts.append(torch.einsum('ctv,tvw->ctw', xr[i, :, :, :p], A[i, :, :p, :p]))
        gy = torch.randn(n, c, 8, v, generator=g)
        loss = sum((parts[i] * gy[i, :, :, :p]).sum() for i, p in enumerate(peds))
        loss.backward()
        for i, p in enumerate(peds):
            ref[i, :, :, :p] = parts[i].detach()
    else:
        ref_t = torch.einsum(eq, xr, A)
        gy = torch.randn(n, c, 8, v, generator=g)
        (ref_t * gy).sum().backward()
        ref = ref_t.detach()
    xd = x.to(dev).requires_grad_(True)
    y = ops.spatial_agg(xd, A.to(dev), peds)
    (y * gy.to(dev)).sum().backward()
    scale = max(1.0, float(ref.abs().max()))
    assert _maxdiff(y.detach().cpu().numpy(), ref.numpy()) < 2e-5 * scale
    gref = xr.grad.numpy()
    assert _maxdiff(xd.grad.cpu().numpy(), gref) < 2e-5 * max(1.0, np.abs(gref).max())


@pytest.mark.parametrize("cin,cout,kt,pad,v", [(2, 5, 1, 0, 9), (5, 5, 3, 1, 32), (3, 7, 3, 0, 5)])
def test_conv_t(dev, cin, cout, kt, pad, v):
    from social_stgcnn_amd import ops
    g = torch.Generator().manual_seed(cin * 10 + kt)
    x = torch.randn(3, cin, 8, v, generator=g)
    w = torch.randn(cout, cin, kt, 1, generator=g) * 0.3
    b = torch.randn(cout, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.nn.functional.conv2d(xr, wr, br, padding=(pad, 0))
    gy = torch.randn(ref.shape, generator=g)
    (ref * gy).sum().backward()
    xd, wd, bd = [t.to(dev).requires_grad_(True) for t in (x, w, b)]
    y = ops.conv_t(xd, wd, bd, pad)
    (y * gy.to(dev)).sum().backward()
    assert _maxdiff(y.detach().cpu().numpy(), ref.detach().numpy()) < 2e-5
    assert _maxdiff(xd.grad.cpu().numpy(), xr.grad.numpy()) < 2e-5
    assert _maxdiff(wd.grad.cpu().numpy(), wr.grad.numpy()) < 1e-4
    assert _maxdiff(bd.grad.cpu().numpy(), br.grad.numpy()) < 1e-4


# ------------------------------------------------------------------------------------------
def test_bivariate_loss_golden(dev):
    """R6 value and gradient against the reference fixture (incl. clamp-active element)."""
    from social_stgcnn_amd.metrics import bivariate_loss
    g = load_golden("loss_cases.npz")
    # hand the kernel the same strided view the training loop produces (train.py:52)
    vp_store = torch.from_numpy(g["vpred"]).permute(2, 0, 1).contiguous().to(dev)       # (5,P,V)
    vp_store.requires_grad_(True)
    vp = vp_store.permute(1, 2, 0)                                                     # (P,V,5) view
    loss = bivariate_loss(vp, torch.from_numpy(g["vtrgt"]).to(dev))
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 2e-6
    grad = vp_store.grad.permute(1, 2, 0).cpu().numpy()
    assert _maxdiff(grad, g["dvpred"]) < 2e-6 * max(1.0, np.abs(g["dvpred"]).max())
    assert np.all(grad[0, 0] == 0)


def test_bivariate_loss_propagates_nan_like_torch(dev):
    """|corr logit| > 10: tanh saturates to rho = +-1, 1 - rho^2 = 0 and the pdf is 0/0.  torch.clamp does not clamp a
    NaN, so the reference's loss and that element's five gradients are NaN (a diverged run is visible); every other
    element keeps its finite gradient.  Checked against the oracle (metrics.py:84-113 restated on torch ops)."""
    from social_stgcnn_amd.metrics import bivariate_loss
    O = _oracle()
    gen = torch.Generator().manual_seed(3)
    vp = torch.randn(12, 4, 5, generator=gen) * 0.3
    vt = torch.randn(12, 4, 2, generator=gen)
    vp[0, 0, 4], vp[5, 2, 4] = 20.0, -20.0
    ref = vp.clone().requires_grad_(True)
    l_ref = O.bivariate_loss(ref, vt)
    l_ref.backward()
    assert torch.isnan(l_ref)
    got = vp.clone().to(dev).requires_grad_(True)
    loss = bivariate_loss(got, vt.to(dev))
    loss.backward()
    assert torch.isnan(loss)
    g, r = got.grad.cpu().numpy(), ref.grad.numpy()
    assert np.array_equal(np.isnan(g), np.isnan(r)) and np.isnan(g).sum() == 10
    ok = ~np.isnan(r)
    assert np.abs(g[ok] - r[ok]).max() < 2e-6 * max(1.0, np.abs(r[ok]).max())


def test_bivariate_loss_batched(dev):
    from social_stgcnn_amd.metrics import bivariate_loss
    O = _oracle()
    g = torch.Generator().manual_seed(5)
    n, p, v = 4, 12, 9
    peds = [9, 4, 1, 7]
    vp = torch.randn(n, p, v, 5, generator=g) * 0.5
    vt = torch.randn(n, p, v, 2, generator=g)
    vpr = vp.clone().requires_grad_(True)
    ref = torch.stack([O.bivariate_loss(vpr[i, :, :k], vt[i, :, :k]) for i, k in enumerate(peds)])
    wts = torch.tensor([0.3, 1.0, 2.0, -0.5])
    (ref * wts).sum().backward()
    vpd = vp.to(dev).requires_grad_(True)
    out = bivariate_loss(vpd, vt.to(dev), peds)
    (out * wts.to(dev)).sum().backward()
    assert _maxdiff(out.detach().cpu().numpy(), ref.detach().numpy()) < 2e-6
    assert _maxdiff(vpd.grad.cpu().numpy(), vpr.grad.numpy()) < 2e-6


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("v", VS)
def test_eval_forward_golden(dev, v):
    """R3-R5 with the shipped eth checkpoint: module outputs vs the reference's (fixture).
    Bound: 2e-5 absolute on the bivariate parameters (north-star bar 1e-4)."""
    from social_stgcnn_amd.model import social_stgcnn
    w = _state(load_golden("weights_eth.npz"))
    a = load_golden("adj_cases.npz")
    f = load_golden("forward_eval.npz")
    m = _model(dev, state=w).eval()
    x = torch.from_numpy(a["nodes_%d" % v]).unsqueeze(0).permute(0, 3, 1, 2).to(dev)    # strided view
    A = torch.from_numpy(a["lap_%d" % v]).to(dev)
    with torch.no_grad():
        y, a_out = m(x, A)
        g, _ = m.st_gcns[0].gcn(x, A)
        h, _ = m.st_gcns[0](x, A)
        y2, _ = m(x, A)                 # parent re-packs after the child call
    assert a_out is A
    assert y.shape == (1, 5, 12, v) and y.is_contiguous()
    assert _maxdiff(g.cpu().numpy(), f["gcn_%d" % v]) < 1e-5
    assert _maxdiff(h.cpu().numpy(), f["stgcn_%d" % v]) < 1e-5
    assert _maxdiff(y.cpu().numpy(), f["vpred_%d" % v]) < 2e-5
    assert torch.equal(y, y2)


def _train_case(dev, v, waves=None):
    t = load_golden("train_fwd_bwd.npz")
    a = load_golden("adj_cases.npz")
    state = _state(t, "sd_%d/" % v)
    m = _model(dev, state=state).train()
    x = torch.from_numpy(a["nodes_%d" % v]).unsqueeze(0).permute(0, 3, 1, 2).to(dev)
    A = torch.from_numpy(a["lap_%d" % v]).to(dev)
    tgt = torch.from_numpy(a["prednodes_%d" % v]).to(dev)
    return t, m, x, A, tgt


@pytest.mark.parametrize("v", (3, 17, 57))
def test_train_forward_backward_golden(dev, v):
    """Reference train-mode forward + bivariate_loss + backward (fixture): V_pred, loss, every
    parameter gradient (dead ones stay None), BatchNorm buffers after the step."""
    from social_stgcnn_amd.metrics import bivariate_loss
    t, m, x, A, tgt = _train_case(dev, v)
    y, _ = m(x, A)
    y.retain_grad()
    v_pred = y.permute(0, 2, 3, 1).squeeze(0)
    loss = bivariate_loss(v_pred, tgt)
    loss.backward()
    assert _maxdiff(y.detach().cpu().numpy(), t["vpred_%d" % v]) < 2e-5
    assert abs(loss.item() - float(t["loss_%d" % v])) < 2e-5
    assert _maxdiff(y.grad.cpu().numpy(), t["dvpred_%d" % v]) < 1e-6
    def ref_of(name):
        ref = t["grad_%d/%s" % (v, name)]
        return None if np.isnan(ref).all() else ref
    worst = _grad_errors(((name, p.grad) for name, p in m.named_parameters()), ref_of)
    bad = {k: e for k, e in worst.items() if e > 2e-4}
    assert not bad, "relative gradient errors: %s" % bad
    for k, val in m.state_dict().items():
        if "running" in k:
            assert _maxdiff(val.cpu().numpy(), t["after_%d/%s" % (v, k)]) < 1e-6, k
        if "num_batches" in k:
            assert int(val) == int(t["after_%d/%s" % (v, k)]), k


def test_batched_ragged_train_step_vs_oracle(dev):
    """A padded ragged batch (num_peds) in one launch == the oracle's per-scene N=1 loop:
    per-scene V_pred and losses, summed parameter gradients, sequentially folded BN buffers."""
    from social_stgcnn_amd.metrics import bivariate_loss
    O = _oracle()
    a = load_golden("adj_cases.npz")
    vs = [5, 17, 2, 32, 8, 3, 57, 17]
    vmax = 57
    m = _model(dev, seed=21).train()
    state = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    keys = [k for k, _ in m.named_parameters()]
    n = len(vs)
    x = torch.zeros(n, 2, 8, vmax)
    A = torch.zeros(n, 8, vmax, vmax)
    tgt = torch.zeros(n, 12, vmax, 2)
    for i, v in enumerate(vs):
        x[i, :, :, :v] = torch.from_numpy(a["nodes_%d" % v]).permute(2, 0, 1)
        A[i, :, :v, :v] = torch.from_numpy(a["lap_%d" % v])
        tgt[i, :, :v] = torch.from_numpy(a["prednodes_%d" % v])
    wts = torch.linspace(0.5, 1.5, n)
    # oracle: sequential scenes
    params = {k: state[k].clone().requires_grad_(True) for k in keys}
    work = dict(state)
    work.update(params)
    ref_losses, ref_pred = [], []
    for i, v in enumerate(vs):
        l, vp = O.scene_loss(work, x[i:i + 1, :, :, :v], A[i, :, :v, :v], tgt[i, :, :v], True)
        ref_losses.append(l)
        ref_pred.append(vp.detach())
    (torch.stack(ref_losses) * wts).sum().backward()
    # HIP: one batch
    y, _ = m(x.to(dev), A.to(dev), num_peds=vs)
    losses = bivariate_loss(y.permute(0, 2, 3, 1), tgt.to(dev), vs)
    (losses * wts.to(dev)).sum().backward()
    yc = y.detach().cpu()
    for i, v in enumerate(vs):
        assert _maxdiff(yc[i, :, :, :v].permute(1, 2, 0).numpy(), ref_pred[i].numpy()) < 2e-5, i
        assert torch.all(yc[i, :, :, v:] == 0)
    assert _maxdiff(losses.detach().cpu().numpy(), torch.stack(ref_losses).detach().numpy()) < 2e-5
    errs = _grad_errors(((name, p.grad) for name, p in m.named_parameters()),
                        lambda name: None if params[name].grad is None else params[name].grad.numpy())
    bad = {k: e for k, e in errs.items() if e > 2e-4}
    assert not bad, "relative gradient errors: %s" % bad
    for k, val in m.state_dict().items():
        if "running" in k:
            assert _maxdiff(val.cpu().numpy(), work[k].numpy()) < 2e-6, k
        if "num_batches" in k:
            assert int(val) == int(work[k]), k


@pytest.mark.parametrize("wg_path", (False, True))
@pytest.mark.parametrize("waves", (1, 2, 4, 8))
def test_wave_count_variants_agree(dev, waves, wg_path, monkeypatch):
    """Every WAVES instantiation of the workgroup-per-scene kernels (stg_model_desc.wg_waves) computes the same
    scene, on the wave-per-scene path (its st_gcn block kernels) and with the whole model on the workgroup path."""
    from social_stgcnn_amd.metrics import bivariate_loss
    from social_stgcnn_amd import ops
    monkeypatch.setitem(ops.OPTIONS, "wg_waves", waves)
    monkeypatch.setitem(ops.OPTIONS, "wg_path", wg_path)
    t, m, x, A, tgt = _train_case(dev, 17)
    y, _ = m(x, A)
    loss = bivariate_loss(y.permute(0, 2, 3, 1).squeeze(0), tgt)
    loss.backward()
    assert _maxdiff(y.detach().cpu().numpy(), t["vpred_17"]) < 2e-5
    def ref_of(name):
        ref = t["grad_17/%s" % name]
        return None if np.isnan(ref).all() else ref
    errs = _grad_errors(((name, p.grad) for name, p in m.named_parameters()), ref_of)
    bad = {k: e for k, e in errs.items() if e > 2e-4}
    assert not bad, bad


def test_stacked_blocks_and_input_grad(dev):
    """n_stgcnn=2 (identity residual, dx through the first block) and n_txpcnn=2/1 variants,
    including the gradient w.r.t. the input x, against oracle autograd."""
    O = _oracle()
    a = load_golden("adj_cases.npz")
    for n_st, n_tx, v in ((2, 5, 8), (1, 2, 5), (1, 1, 17), (3, 3, 3)):
        m = _model(dev, seed=n_st * 10 + n_tx, n_stgcnn=n_st, n_txpcnn=n_tx).train()
        state = {k: val.detach().cpu().clone() for k, val in m.state_dict().items()}
        keys = [k for k, _ in m.named_parameters()]
        x = torch.from_numpy(a["nodes_%d" % v]).unsqueeze(0).permute(0, 3, 1, 2).contiguous()
        A = torch.from_numpy(a["lap_%d" % v])
        params = {k: state[k].clone().requires_grad_(True) for k in keys}
        work = dict(state)
        work.update(params)
        xr = x.clone().requires_grad_(True)
        ref = O.social_stgcnn_forward(work, xr, A, True, n_stgcnn=n_st, n_txpcnn=n_tx)
        g = torch.Generator().manual_seed(3)
        gy = torch.randn(ref.shape, generator=g)
        (ref * gy).sum().backward()
        xd = x.to(dev).requires_grad_(True)
        y, _ = m(xd, A.to(dev))
        (y * gy.to(dev)).sum().backward()
        assert _maxdiff(y.detach().cpu().numpy(), ref.detach().numpy()) < 2e-5, (n_st, n_tx)
        gs = max(1e-3, float(xr.grad.abs().max()))
        assert _maxdiff(xd.grad.cpu().numpy(), xr.grad.numpy()) / gs < 2e-4, (n_st, n_tx)
        errs = _grad_errors(((name, p.grad) for name, p in m.named_parameters()),
                            lambda name: None if params[name].grad is None else params[name].grad.numpy())
        bad = {k: e for k, e in errs.items() if e > 2e-4}
        assert not bad, (n_st, n_tx, bad)


def test_errors_are_loud(dev):
    from social_stgcnn_amd import _lib, ops
    from social_stgcnn_amd.model import social_stgcnn, st_gcn
    with pytest.raises(RuntimeError):
        ops.spatial_agg(torch.zeros(1, 2, 8, 4), torch.zeros(8, 4, 4))          # CPU tensor: no fallback
    m = social_stgcnn(n_stgcnn=1, n_txpcnn=5, seq_len=8, pred_seq_len=12).to(dev)
    with pytest.raises(AssertionError):
        m(torch.zeros(1, 2, 8, 4, device=dev), torch.zeros(7, 4, 4, device=dev))   # model.py:65 assert
    with pytest.raises(RuntimeError):
        social_stgcnn(n_stgcnn=1, n_txpcnn=5, seq_len=6, pred_seq_len=12).to(dev)(
            torch.zeros(1, 2, 6, 4, device=dev), torch.zeros(6, 4, 4, device=dev))   # unsupported seq_len
    with pytest.raises(AssertionError):
        st_gcn(2, 5, (2, 8))                                                       # model.py:105 assert
    assert _lib.lib().stg_adj_build(None, 0, 0, 0, 0, None, 1, 4, 8, 1, None, None, None) == -1
    assert b"null" in _lib.lib().stg_last_error()


def test_full_size_properties(dev):
    """BASELINE sizes (N=2048, V=32): size-independent properties instead of a CPU oracle run:
    scene permutation equivariance, determinism, gradient additivity over batch halves."""
    from social_stgcnn_amd import ops
    from social_stgcnn_amd.metrics import bivariate_loss
    rng = np.random.default_rng(1)
    n, v = 2048, 32
    rel = torch.from_numpy(rng.uniform(-0.5, 0.5, (n, v, 2, 8)).round(4).astype(np.float32)).to(dev)
    rel[:, :, :, 0] = 0
    nodes, A = ops.adj_build(rel)
    x = nodes.permute(0, 3, 1, 2)
    tgt = torch.from_numpy(rng.uniform(-0.5, 0.5, (n, 12, v, 2)).astype(np.float32)).to(dev)
    m = _model(dev, seed=0).train()

    def step(idx):
        m.zero_grad(set_to_none=True)
        y, _ = m(x[idx], A[idx])
        l = bivariate_loss(y.permute(0, 2, 3, 1), tgt[idx])
        l.sum().backward()
        return y.detach(), l.detach(), torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None])

    all_idx = torch.arange(n, device=dev)
    y0, l0, g0 = step(all_idx)
    y1, l1, g1 = step(all_idx)
    assert torch.equal(y0, y1) and torch.equal(l0, l1)                               # deterministic
    assert float((g1 - g0).abs().max()) < 1e-5 * float(g0.abs().max())               # LDS-atomic order only
    perm = torch.from_numpy(rng.permutation(n)).to(dev)
    yp, lp, gp = step(perm)
    assert torch.equal(yp, y0[perm]) and torch.equal(lp, l0[perm])                   # equivariance
    assert float((gp - g0).abs().max()) < 1e-4 * float(g0.abs().max())               # order-only noise
    _, _, ga = step(all_idx[: n // 2])
    _, _, gb = step(all_idx[n // 2:])
    assert float((ga + gb - g0).abs().max()) < 1e-4 * float(g0.abs().max())          # additivity
    assert torch.isfinite(y0).all() and torch.isfinite(g0).all()
    # symmetric Laplacian, zero first frame, rows of D^-1/2 (D-A) D^-1/2 in [-1, 1]
    assert torch.equal(A, A.transpose(2, 3)) and torch.all(A[:, 0] == 0) and float(A.abs().max()) <= 1.0


def _synthetic_scene(v, seed):
    rng = np.random.default_rng(seed)
    rel = np.zeros((v, 2, 20), np.float32)
    rel[:, :, 1:] = np.round(rng.uniform(-0.6, 0.6, (v, 2, 19)), 4).astype(np.float32)
    return rel


@pytest.mark.parametrize("v,force_generic", [(128, False), (96, False), (17, True), (57, True), (68, False), (69, False)])
def test_large_v_and_workgroup_path(dev, v, force_generic, monkeypatch):
    """cfg5-style dense crowds (V=128 needs the workgroup-per-scene kernels: 8 waves, ~110 KB of LDS) and the
    workgroup-per-scene path forced on small scenes (ops.OPTIONS["wg_path"]) -- forward, loss, every gradient and
    the BatchNorm buffers against the oracle; 68 / 69 straddle the limit of the wave-per-scene path."""
    from social_stgcnn_amd import ops
    from social_stgcnn_amd.metrics import bivariate_loss
    O = _oracle()
    if force_generic:
        monkeypatch.setitem(ops.OPTIONS, "wg_path", True)
    n = 3
    rels = [_synthetic_scene(v, 100 + v + i) for i in range(n)]
    m = _model(dev, seed=v).train()
    state = {k: val.detach().cpu().clone() for k, val in m.state_dict().items()}
    keys = [k for k, _ in m.named_parameters()]
    # The reference runs in FLOAT64: PReLU's derivative jumps at 0, so a pre-activation within fp32 rounding of
    # zero (about one element in 10^5 scenes-elements) makes two correct fp32 implementations -- even the
    # oracle with 1 vs 4 CPU threads -- disagree on a gradient by O(upstream gradient).  fp64 is the arbiter.
    state64 = {k: (val.double() if val.is_floating_point() else val.clone()) for k, val in state.items()}
    params = {k: state64[k].clone().requires_grad_(True) for k in keys}
    work = dict(state64)
    work.update(params)
    ref_losses, ref_pred = [], []
    for rel in rels:
        nodes, lap = O.seq_to_graph_np(rel[:, :, :8])
        tgt, _ = O.seq_to_graph_np(rel[:, :, 8:])
        l, vp = O.scene_loss(work, torch.from_numpy(nodes).double().unsqueeze(0).permute(0, 3, 1, 2),
                             torch.from_numpy(lap).double(), torch.from_numpy(tgt).double(), True)
        ref_losses.append(l)
        ref_pred.append(vp.detach())
    torch.stack(ref_losses).sum().backward()
    rel_d = torch.from_numpy(np.stack(rels)).to(dev)
    nodes_d, adj_d = ops.adj_build(rel_d[..., :8])
    tgt_d = rel_d[..., 8:].permute(0, 3, 1, 2).contiguous()
    y, _ = m(nodes_d.permute(0, 3, 1, 2), adj_d)
    losses = bivariate_loss(y.permute(0, 2, 3, 1), tgt_d)
    losses.sum().backward()
    for i in range(n):
        assert _maxdiff(y[i].detach().permute(1, 2, 0).cpu().numpy(), ref_pred[i].numpy()) < 5e-5, i
    assert _maxdiff(losses.detach().cpu().numpy(), torch.stack(ref_losses).detach().numpy()) < 5e-5
    errs = _grad_errors(((name, p.grad) for name, p in m.named_parameters()),
                        lambda name: None if params[name].grad is None else params[name].grad.numpy())
    print("V=%d%s: worst relative gradient error %.1e" % (v, " (workgroup path)" if force_generic else "",
                                                          max(errs.values())))
    bad = {k: e for k, e in errs.items() if e > 5e-5}           # measured worst case 3.4e-6 (V = 128)
    assert not bad, bad
    for k, val in m.state_dict().items():
        if "running" in k:
            assert _maxdiff(val.cpu().numpy(), work[k].numpy()) < 2e-6, k


@pytest.mark.auto_path
@pytest.mark.parametrize("n,v", [(3, 12), (200, 12), (700, 12), (200, 30), (500, 30), (3, 57)])
def test_small_batches_are_cut_into_finer_teams_by_themselves(dev, n, v):
    """Default path selection (no option set): a small batch runs the team launch of the exact-bf16 kernels with finer
    class bounds (fewer than 384 scenes: one wave up to 8 pedestrians, two up to 16, four beyond; fewer than 1536: 16 / 32);
    forward, loss, every gradient and the BatchNorm buffers must equal the workgroup-per-scene kernels' result of the same
    batch (a different kernel family: fp32 MFMA, the whole scene in one workgroup)."""
    from social_stgcnn_amd import ops
    from social_stgcnn_amd.metrics import bivariate_loss
    assert not ops.OPTIONS["wave_path"] and not ops.OPTIONS["wg_path"]
    rel = torch.from_numpy(np.stack([_synthetic_scene(v, 7 + i) for i in range(8)])).to(dev)
    rel = rel[torch.arange(n, device=dev) % 8] * (1.0 + 0.001 * torch.arange(n, device=dev)[:, None, None, None])
    peds = torch.tensor([(i * 5) % v + 1 for i in range(n)], dtype=torch.int32, device=dev)
    nodes, adj = ops.adj_build(rel[..., :8], peds)
    tgt = rel[..., 8:].permute(0, 3, 1, 2).contiguous()
    res = []
    for wg in (False, True):
        ops.OPTIONS["wg_path"] = wg
        m = _model(dev, seed=5).train()
        y, _ = m(nodes.permute(0, 3, 1, 2), adj, peds)
        bivariate_loss(y.permute(0, 2, 3, 1), tgt, peds).sum().backward()
        res.append((y.detach(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None},
                    {k: b.clone() for k, b in m.named_buffers()}))
    ops.OPTIONS["wg_path"] = False
    (ya, ga, ba), (yw, gw, bw) = res
    assert float((ya - yw).abs().max()) < 2e-5
    errs = _grad_errors(((k, ga.get(k)) for k in gw), lambda name: gw[name].cpu().numpy())
    bad = {k: e for k, e in errs.items() if e > 1e-4}
    assert not bad, bad
    for k in bw:
        if "running" in k:
            assert float((ba[k] - bw[k]).abs().max()) < 1e-5, k


def test_no_grad_forward_saves_no_activations(dev):
    """A forward under torch.no_grad() (vald(), test()) must run in inference mode -- no activation workspace, no
    saved planes -- even though the parameters require grad; with grad enabled the workspace is there."""
    from social_stgcnn_amd import ops
    m = _model(dev, seed=1).train()
    rel = torch.from_numpy(np.stack([_synthetic_scene(9, 5), _synthetic_scene(9, 6)])).to(dev)
    nodes, adj = ops.adj_build(rel[..., :8])
    x = nodes.permute(0, 3, 1, 2)
    y1, _ = m(x, adj)
    assert m.last_ws_floats > 0
    with torch.no_grad():
        y0, _ = m(x, adj)
    assert m.last_ws_floats == 0
    assert _maxdiff(y0.cpu().numpy(), y1.detach().cpu().numpy()) < 1e-6


def test_inference_matches_training_forward(dev):
    """eval()/no_grad (no activation saves, running statistics) on a ragged batch == per-scene oracle."""
    O = _oracle()
    a = load_golden("adj_cases.npz")
    w = _state(load_golden("weights_eth.npz"))
    m = _model(dev, state=w).eval()
    vs = [3, 32, 8, 57, 2]
    vmax = 57
    x = torch.zeros(len(vs), 2, 8, vmax)
    A = torch.zeros(len(vs), 8, vmax, vmax)
    for i, v in enumerate(vs):
        x[i, :, :, :v] = torch.from_numpy(a["nodes_%d" % v]).permute(2, 0, 1)
        A[i, :, :v, :v] = torch.from_numpy(a["lap_%d" % v])
    with torch.no_grad():
        y, _ = m(x.to(dev), A.to(dev), num_peds=vs)
    f = load_golden("forward_eval.npz")
    for i, v in enumerate(vs):
        assert _maxdiff(y[i, :, :, :v].cpu().numpy(), f["vpred_%d" % v][0]) < 2e-5, v
        assert torch.all(y[i, :, :, v:] == 0)


def test_empty_and_degenerate_batches(dev):
    """Edge cases: an empty batch, scenes with zero pedestrians inside a batch, and single-pedestrian scenes
    (the reference's dataset never emits V=1 windows -- utils.py:171 keeps windows with >1 pedestrians -- but the
    kernels must not misbehave on them)."""
    from social_stgcnn_amd import ops
    from social_stgcnn_amd.metrics import bivariate_loss
    O = _oracle()
    m = _model(dev, seed=5).train()
    state0 = {k: val.detach().cpu().clone() for k, val in m.state_dict().items()}
    # N = 0
    y, _ = m(torch.zeros(0, 2, 8, 4, device=dev), torch.zeros(0, 8, 4, 4, device=dev))
    assert y.shape == (0, 5, 12, 4)
    nodes, adj = ops.adj_build(torch.zeros(0, 4, 2, 8, device=dev))
    assert adj.shape == (0, 8, 4, 4)
    for k, val in m.state_dict().items():
        assert torch.equal(val.cpu(), state0[k]), k                      # nothing moved
    # zero-ped scenes inside a batch: skipped everywhere (outputs 0, no BatchNorm update, no gradient)
    a = load_golden("adj_cases.npz")
    vs = [0, 5, 0, 1, 3]
    vmax = 5
    rel = torch.zeros(len(vs), vmax, 2, 20)
    g = torch.Generator().manual_seed(2)
    for i, v in enumerate(vs):
        rel[i, :v, :, 1:] = (torch.rand(v, 2, 19, generator=g) - 0.5).mul(1e4).round().div(1e4)
    rel_d = rel.to(dev)
    peds = torch.tensor(vs, dtype=torch.int32, device=dev)
    nodes, adj = ops.adj_build(rel_d[..., :8], peds)
    tgt = rel_d[..., 8:].permute(0, 3, 1, 2).contiguous()
    y, _ = m(nodes.permute(0, 3, 1, 2), adj, peds)
    losses = bivariate_loss(y.permute(0, 2, 3, 1), tgt, peds)
    losses.sum().backward()
    assert torch.all(y[0] == 0) and torch.all(y[2] == 0) and float(losses[0].detach()) == 0 and float(losses[2].detach()) == 0
    # oracle on the three non-empty scenes (V = 5, 1, 3)
    keys = [k for k, _ in m.named_parameters()]
    params = {k: state0[k].clone().requires_grad_(True) for k in keys}
    work = dict(state0)
    work.update(params)
    tot = 0
    for i, v in enumerate(vs):
        if v == 0:
            continue
        n_i, l_i = O.seq_to_graph_np(rel[i, :v, :, :8].numpy())
        t_i, _ = O.seq_to_graph_np(rel[i, :v, :, 8:].numpy())
        l, vp = O.scene_loss(work, torch.from_numpy(n_i).unsqueeze(0).permute(0, 3, 1, 2), torch.from_numpy(l_i),
                             torch.from_numpy(t_i), True)
        tot = tot + l
        assert _maxdiff(y[i, :, :, :v].detach().permute(1, 2, 0).cpu().numpy(), vp.detach().numpy()) < 2e-5, i
        assert abs(float(losses[i]) - float(l)) < 2e-5
    tot.backward()
    errs = _grad_errors(((name, p.grad) for name, p in m.named_parameters()),
                        lambda name: None if params[name].grad is None else params[name].grad.numpy())
    bad = {k: e for k, e in errs.items() if e > 5e-5}           # measured worst case 3.4e-6 (V = 128)
    assert not bad, bad
    for k, val in m.state_dict().items():
        if "running" in k:
            assert _maxdiff(val.cpu().numpy(), work[k].numpy()) < 2e-6, k
        if "num_batches" in k:
            assert int(val) == 3, k                                         # only the non-empty scenes count


@pytest.mark.parametrize("n,v", [(1, 5), (2, 3), (70, 5), (2048, 57), (5000, 200), (16384, 1023)])
def test_scene_order_is_a_stable_descending_sort(dev, n, v):
    """The ragged-batch schedule: stg_scene_order == stable argsort of the clamped pedestrian counts, descending."""
    from social_stgcnn_amd import ops
    g = torch.Generator().manual_seed(n + v)
    peds = torch.randint(-2, v + 3, (n,), generator=g, dtype=torch.int32)
    got, key_start = ops.scene_order(peds.to(dev), v)
    got, key_start = got.cpu(), key_start.cpu()
    key = peds.clamp(0, v)
    want = torch.sort(key, descending=True, stable=True).indices.to(torch.int32)
    assert torch.equal(got, want)
    for k in range(v + 2):                       # key_start[k] = #scenes with more than v-k pedestrians
        assert int(key_start[k]) == int((key > v - k).sum()), k


def test_ragged_batch_is_order_invariant(dev):
    """A shuffled ragged batch (internally sorted by crowd size and dealt boustrophedon to the persistent waves)
    gives per-scene outputs identical to running every scene alone, and gradients equal to their sum."""
    import bench
    from social_stgcnn_amd import ops
    counts = bench.ragged_counts(300, seed=3)
    v = int(counts.max())
    obs_rel, target = bench.synth_scenes(300, v, seed=3)
    live = np.arange(v)[None, :] < counts[:, None]
    obs_rel *= live[:, :, None, None]
    target *= live[:, None, :, None]
    peds = torch.from_numpy(counts).to(dev)
    nodes, adj = ops.adj_build(torch.from_numpy(obs_rel).to(dev), peds)
    x, tgt = nodes.permute(0, 3, 1, 2), torch.from_numpy(target).to(dev)
    m = _model(dev, seed=5).train()
    y, _ = m(x, adj, peds)
    w = torch.linspace(0.5, 1.5, 300, device=dev)
    losses, dy = ops.bivariate_nll_with_grad(y.detach(), tgt, peds, w)
    y.backward(dy)
    g_batch = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    # the same scenes one at a time, compact (unpadded) tensors, no ordering involved
    m2 = _model(dev, seed=5).train()
    for i in range(0, 300, 7):
        c = int(counts[i])
        yi, _ = m2(x[i:i + 1, :, :, :c], adj[i:i + 1, :, :c, :c])
        assert float((yi[0] - y[i, :, :, :c]).abs().max()) < 2e-5, i
        assert torch.all(y[i, :, :, c:] == 0)
        li, dyi = ops.bivariate_nll_with_grad(yi.detach(), tgt[i:i + 1, :, :c].contiguous(), None, w[i:i + 1])
        assert abs(float(li) - float(losses[i])) < 1e-6
    # gradient of the whole batch == gradient with the scenes presented in sorted order (pure permutation)
    perm = torch.argsort(peds, descending=True, stable=True)
    m3 = _model(dev, seed=5).train()
    y3, _ = m3(x[perm], adj[perm], peds[perm])
    _, dy3 = ops.bivariate_nll_with_grad(y3.detach(), tgt[perm], peds[perm], w[perm])
    y3.backward(dy3)
    for k, p in m3.named_parameters():
        if p.grad is None:
            continue
        a, b = g_batch[k], p.grad
        assert float((a - b).abs().max()) <= 2e-5 * max(1e-3, float(b.abs().max())), k


def test_every_crowd_size_of_the_wave_path(dev):
    """V = 1 .. 70, one launch per V holding a full scene and a ragged one (V_n = V - 1, padded): forward, loss and
    every gradient against the fp64 oracle.  The LDS layouts pad channel strides to 16 (mod 32) in three different
    geometries (two-plane, in-place, position-major); this walks every padding case of each."""
    from social_stgcnn_amd import ops
    from social_stgcnn_amd.metrics import bivariate_loss
    O = _oracle()
    m = _model(dev, seed=77).train()
    state = {k: val.detach().cpu().clone() for k, val in m.state_dict().items()}
    keys = [k for k, _ in m.named_parameters()]
    state64 = {k: (val.double() if val.is_floating_point() else val.clone()) for k, val in state.items()}
    worst = {}
    for v in range(1, 71):
        counts = [v, max(1, v - 1)]
        rels = [_synthetic_scene(v, 1000 + 2 * v), _synthetic_scene(v, 1001 + 2 * v)]
        rels[1][counts[1]:] = 0.0
        params = {k: state64[k].clone().requires_grad_(True) for k in keys}
        work = {k: (val.clone() if torch.is_tensor(val) else val) for k, val in state64.items()}
        work.update(params)
        ref_losses, ref_pred = [], []
        for rel, c in zip(rels, counts):
            nodes, lap = O.seq_to_graph_np(rel[:c, :, :8])
            tgt, _ = O.seq_to_graph_np(rel[:c, :, 8:])
            l, vp = O.scene_loss(work, torch.from_numpy(nodes).double().unsqueeze(0).permute(0, 3, 1, 2),
                                 torch.from_numpy(lap).double(), torch.from_numpy(tgt).double(), True)
            ref_losses.append(l)
            ref_pred.append(vp.detach())
        torch.stack(ref_losses).sum().backward()
        with torch.no_grad():
            m.load_state_dict(state)
        for p in m.parameters():
            p.grad = None
        rel_d = torch.from_numpy(np.stack(rels)).to(dev)
        peds = torch.tensor(counts, dtype=torch.int32, device=dev)
        nodes_d, adj_d = ops.adj_build(rel_d[..., :8], peds)
        tgt_d = rel_d[..., 8:].permute(0, 3, 1, 2).contiguous()
        y, _ = m(nodes_d.permute(0, 3, 1, 2), adj_d, peds)
        losses = bivariate_loss(y.permute(0, 2, 3, 1), tgt_d, peds)
        losses.sum().backward()
        for i, c in enumerate(counts):
            err = _maxdiff(y[i, :, :, :c].detach().permute(1, 2, 0).cpu().numpy(), ref_pred[i].numpy())
            assert err < 5e-5, (v, i, err)
            assert torch.all(y[i, :, :, c:] == 0), (v, i)
        assert _maxdiff(losses.detach().cpu().numpy(), torch.stack(ref_losses).detach().numpy()) < 5e-5, v
        errs = _grad_errors(((name, p.grad) for name, p in m.named_parameters()),
                            lambda name: None if params[name].grad is None else params[name].grad.numpy())
        bad = {k: e for k, e in errs.items() if e > 1e-4}       # measured worst case 2.4e-5 (V = 1)
        assert not bad, (v, bad)
        worst[v] = max(errs.values())
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    print("crowd-size sweep 1..70, worst relative gradient error vs the fp64 oracle:",
          ", ".join("V=%d %.1e" % kv for kv in top))
    assert max(worst.values()) < 1e-4


@pytest.mark.parametrize("counts", ([57, 2, 33, 17, 8, 32, 5, 40], [32, 2, 31, 17, 8, 32, 5, 24]),
                         ids=("crowds_to_57_fp32_mfma_chains", "crowds_to_32_bf16_pipe_kernels"))
def test_bf16_storage_mode_measured_error(dev, monkeypatch, counts):
    """BASELINE configs[2]: STG_OPT_BF16_STORE keeps the saved TXP planes a_l, the pre-activations z_l and the dz_l
    hand-off in bf16 (fp32 compute, accumulation, parameters, inputs).  What that costs, measured against the fp32
    oracle on a ragged batch (2..57 pedestrians): V_pred and the loss are UNCHANGED (the forward computes from its
    fp32 LDS images: the north-star's 1e-4 bar on the five Gaussian parameters holds); the st_gcn block gradients stay at
    fp32 accuracy (the input-gradient chain only needs the sign of z); the TXP weight / bias and PReLU slope gradients
    carry the bf16 rounding of a_l / dz_l / z_l: up to ~1e-3 of the tensor's largest entry."""
    from social_stgcnn_amd import ops
    from social_stgcnn_amd.metrics import bivariate_loss
    O = _oracle()
    monkeypatch.setitem(ops.OPTIONS, "bf16_store", True)
    vmax = max(counts)
    rels = [_synthetic_scene(vmax, 300 + i) for i in range(len(counts))]
    for r, c in zip(rels, counts):
        r[c:] = 0.0
    m = _model(dev, seed=11).train()
    state = {k: val.detach().cpu().clone() for k, val in m.state_dict().items()}
    keys = [k for k, _ in m.named_parameters()]
    params = {k: state[k].clone().requires_grad_(True) for k in keys}
    work = dict(state)
    work.update(params)
    ref_losses, ref_pred = [], []
    for rel, c in zip(rels, counts):
        nodes, lap = O.seq_to_graph_np(rel[:c, :, :8])
        tgt, _ = O.seq_to_graph_np(rel[:c, :, 8:])
        l, vp = O.scene_loss(work, torch.from_numpy(nodes).unsqueeze(0).permute(0, 3, 1, 2), torch.from_numpy(lap),
                             torch.from_numpy(tgt), True)
        ref_losses.append(l)
        ref_pred.append(vp.detach())
    torch.stack(ref_losses).sum().backward()
    rel_d = torch.from_numpy(np.stack(rels)).to(dev)
    peds = torch.tensor(counts, dtype=torch.int32, device=dev)
    nodes_d, adj_d = ops.adj_build(rel_d[..., :8], peds)
    y, _ = m(nodes_d.permute(0, 3, 1, 2), adj_d, peds)
    losses = bivariate_loss(y.permute(0, 2, 3, 1), rel_d[..., 8:].permute(0, 3, 1, 2).contiguous(), peds)
    losses.sum().backward()
    for i, c in enumerate(counts):
        assert _maxdiff(y[i, :, :, :c].detach().permute(1, 2, 0).cpu().numpy(), ref_pred[i].numpy()) < 2e-5
    assert _maxdiff(losses.detach().cpu().numpy(), torch.stack(ref_losses).detach().numpy()) < 2e-5
    errs = _grad_errors(((name, p.grad) for name, p in m.named_parameters()),
                        lambda name: None if params[name].grad is None else params[name].grad.numpy())
    exact = {k: e for k, e in errs.items() if k.startswith("st_gcns")}
    rounded = {k: e for k, e in errs.items() if k not in exact}
    print("bf16 storage: V_pred / loss unchanged; worst relative gradient error: st_gcn block %.1e, TXP weights / "
          "biases / PReLU slopes %.1e (%s)" % (max(exact.values()), max(rounded.values()), max(rounded, key=rounded.get)))
    assert max(exact.values()) < 1e-5, exact               # measured 3e-7
    assert max(rounded.values()) < 3e-3, rounded           # measured 7.8e-4 (prelus.1.weight): bf16 keeps 8 bits


def test_split_bf16_input_gradient_variant(dev, monkeypatch):
    """ops.OPTIONS["split_bf16"] (opt-in, STG_OPT_SPLIT_BF16): the input-gradient GEMMs on bf16 MFMAs with hi/lo-split operands
    (hi*hi + hi*lo + lo*hi, fp32 accumulate).  Same checks, same tolerances as the fp32 path: the ragged batch against
    the oracle, the golden forward/backward cases, and a V sweep subset against the fp64 oracle."""
    from social_stgcnn_amd import ops
    monkeypatch.setitem(ops.OPTIONS, "split_bf16", True)
    test_batched_ragged_train_step_vs_oracle(dev)
    for v in (3, 17, 57):
        test_train_forward_backward_golden(dev, v)
    test_ragged_batch_is_order_invariant(dev)


def test_bf16_pipe_kernels_agree_with_the_fp32_mfma_kernels(dev, monkeypatch):
    """The default convolution / weight-gradient kernels (v_mfma_f32_16x16x32_bf16, exact three-piece operands) against
    the fp32-MFMA kernels (STG_OPT_F32_MFMA) on one ragged 96-scene batch: same accuracy class -- V_pred and every
    gradient agree to a few fp32 roundings of the accumulated sums."""
    import bench
    from social_stgcnn_amd import ops
    from social_stgcnn_amd.model import social_stgcnn
    from social_stgcnn_amd.trainer import Trainer
    n, v = 96, 32
    obs_rel, target = bench.synth_scenes(n, v, 44)
    nodes, adj = ops.adj_build(torch.from_numpy(obs_rel).to(dev))
    x, tgt = nodes.permute(0, 3, 1, 2), torch.from_numpy(target).to(dev)
    peds = torch.randint(1, v + 1, (n,), generator=torch.Generator().manual_seed(3)).to(torch.int32)
    peds[:8] = v
    peds = peds.to(dev)
    w = torch.full((n,), 1.0 / n, device=dev)
    out = {}
    for f32 in (False, True):
        monkeypatch.setitem(ops.OPTIONS, "f32_mfma", f32)
        torch.manual_seed(21)
        m = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12).to(dev).train()
        tr = Trainer(m, lr=0.01)
        total, losses, y = tr.forward_backward(x, adj, tgt, peds, w)
        out[f32] = (y.cpu(), losses.cpu(), {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters() if p.grad is not None})
    ya, la, ga = out[False]
    yb, lb, gb = out[True]
    assert float((ya - yb).abs().max()) < 5e-6 * max(1.0, float(yb.abs().max()))
    assert torch.allclose(la, lb, rtol=1e-5, atol=1e-6)
    gmax = max(float(g.abs().max()) for g in gb.values())
    for k, g in gb.items():
        scale = max(0.05 * gmax, float(g.abs().max()))
        assert float((ga[k] - g).abs().max()) <= 2e-5 * scale, (k, float((ga[k] - g).abs().max()), scale)


def test_kernel_options_belong_to_a_model(dev):
    """ops.KernelOptions: launch options are per model, not process state -- two models in one process, one with its own
    options (the fp32-MFMA kernels, whose training workspace carries no prepared bf16 operands), run their own kernel
    families side by side and agree; the process-wide defaults are untouched."""
    from social_stgcnn_amd import ops
    with pytest.raises(KeyError):
        ops.KernelOptions(no_such_option=True)
    before = dict(ops.OPTIONS)
    rel = torch.from_numpy(np.stack([_synthetic_scene(20, 40 + i) for i in range(6)])).to(dev)
    nodes, adj = ops.adj_build(rel[..., :8])
    x = nodes.permute(0, 3, 1, 2)
    m1, m2 = _model(dev, seed=3).train(), _model(dev, seed=3).train()
    m2.options = ops.KernelOptions(f32_mfma=True, wave_path=True)
    y1, _ = m1(x, adj)
    y2, _ = m2(x, adj)
    y1b, _ = m1(x, adj)
    assert m1.last_ws_floats - m2.last_ws_floats == 5 * 24 * 64 * 4          # the operand tables of five layers
    assert torch.equal(y1, y1b) and _maxdiff(y1.detach().cpu().numpy(), y2.detach().cpu().numpy()) < 2e-5
    assert dict(ops.OPTIONS) == before
