"""GPU (-m gpu): the rows SURVEY 8(f) ranks next -- N2 evaluation tail as one device op (stg_bestofk_eval),
N3 fused clip + SGD + scheduled lr (stg_optim_step) -- against the oracle / torch's own clip_grad_norm_."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from test_gpu_trainer import _eth_batcher, _state

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", 0)


def test_best_of_k_equals_oracle_given_the_same_draws(dev):
    """Ragged batch, strided V_pred, K=20: per-ped min ADE / FDE == oracle.best_of_k_errors_noise (tolerance 1e-4:
    fp32 cumulative sums on the device vs fp64 error norms in the oracle)."""
    from oracle import stgcnn_oracle as o
    from social_stgcnn_amd import ops
    n, p, v, k = 6, 12, 9, 20
    g = torch.Generator().manual_seed(5)
    y = torch.randn(n, p, v, 5, generator=g) * 0.6                 # (N,P,V,5): permuted view -> strided (N,5,P,V)
    tgt = torch.randn(n, p, v, 2, generator=g) * 0.4
    obs = torch.randn(n, v, 2, generator=g) * 6
    eps = torch.randn(k, n, p, v, 2, generator=g)
    peds = torch.tensor([9, 1, 4, 0, 7, 9], dtype=torch.int32)
    a, f = ops.best_of_k(y.to(dev).permute(0, 3, 1, 2), tgt.to(dev), obs.to(dev), peds.to(dev), k, eps.to(dev))
    a, f = a.cpu().numpy(), f.cpu().numpy()
    for i in range(n):
        c = int(peds[i])
        assert np.all(a[i, c:] == 0) and np.all(f[i, c:] == 0)
        if c == 0:
            continue
        ra, rf = o.best_of_k_errors_noise(y[i, :, :c], obs[i, :c].numpy(), tgt[i, :, :c].numpy(), eps[:, i, :, :c])
        np.testing.assert_allclose(a[i, :c], ra, rtol=0, atol=1e-4)
        np.testing.assert_allclose(f[i, :c], rf, rtol=0, atol=1e-4)
    # obs_last is optional (errors are translation invariant up to rounding)
    a2, f2 = ops.best_of_k(y.to(dev).permute(0, 3, 1, 2), tgt.to(dev), None, peds.to(dev), k, eps.to(dev))
    np.testing.assert_allclose(a2.cpu().numpy(), a, rtol=0, atol=1e-4)
    np.testing.assert_allclose(f2.cpu().numpy(), f, rtol=0, atol=1e-4)


def test_best_of_k_philox_stream_statistics(dev):
    """In-kernel draws: deterministic per seed, different across seeds, standard-normal moments, and the
    Cholesky factor applied (correlation shows up in the spread along the diagonal)."""
    from social_stgcnn_amd import ops
    n, v = 4096, 32
    y = torch.zeros(n, 5, 1, v, device=dev)                       # mean 0, sx = sy = 1, rho = 0
    tgt = torch.zeros(n, 1, v, 2, device=dev)
    a0, f0 = ops.best_of_k(y, tgt, None, None, 1, None, seed=7)
    a1, _ = ops.best_of_k(y, tgt, None, None, 1, None, seed=7)
    a2, _ = ops.best_of_k(y, tgt, None, None, 1, None, seed=8)
    assert torch.equal(a0, a1) and torch.equal(a0, f0) and not torch.equal(a0, a2)
    r = a0.double().flatten()                                     # |z|, z ~ N(0, I2): Rayleigh
    assert abs(float(r.mean()) - np.sqrt(np.pi / 2)) < 0.01
    assert abs(float((r * r).mean()) - 2.0) < 0.02
    # best of K decreases with K
    a20, _ = ops.best_of_k(y, tgt, None, None, 20, None, seed=7)
    assert float(a20.mean()) < 0.5 * float(a0.mean())
    # correlation: far-away target along (1,1): error ~ c*sqrt(2) + (dx+dy)/sqrt(2), var = (sx^2+sy^2+2 rho sx sy)/2
    rho_raw, la, lb = 0.5, np.log(1.5), np.log(0.5)
    y[:, 2], y[:, 3], y[:, 4] = la, lb, rho_raw
    tgt[...] = -200.0
    d, _ = ops.best_of_k(y, tgt, None, None, 1, None, seed=3)
    var = float(d.double().var())
    want = (1.5 ** 2 + 0.5 ** 2 + 2 * np.tanh(rho_raw) * 1.5 * 0.5) / 2
    assert abs(var - want) < 0.03 * want, (var, want)


def _eth_eval_batches(dev, n_sc=70, bs=32):
    batcher, pos, rel, peds = _eth_batcher(dev, n_sc)
    batches = []
    for lo in range(0, n_sc, bs):
        hi = min(n_sc, lo + bs)
        x, adj, _, pd = batcher(lo, hi)
        batches.append((x, adj, pd, pos[lo:hi, :, :, 7], np.transpose(rel[lo:hi, :, :, 8:], (0, 3, 1, 2))))
    return batches, peds


def test_device_evaluation_reproduces_reference_test_ade_fde(dev):
    """R10 through the device op: the noise handed to the kernel is the reference sampler's own stream
    (torch.manual_seed(0); per scene, per sample, one (P, V_i, 2) standard-normal draw -- what
    MultivariateNormal.sample() consumes in test.py:89), so the golden per-ped ADE / FDE of the reference's test()
    must come back to 1e-4."""
    from social_stgcnn_amd.model import social_stgcnn
    from social_stgcnn_amd.trainer import evaluate_ade_fde_device
    g = load_golden("eval_ade_fde.npz")
    m = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12)
    m.load_state_dict(_state(load_golden("weights_eth.npz")))
    m.to(dev)
    batches, peds = _eth_eval_batches(dev)
    torch.manual_seed(0)
    lo_of = [0]

    def noise_fn(b, shape):
        k, n, p, v, _ = shape
        out = torch.zeros(shape)
        for i in range(n):
            c = int(peds[lo_of[0] + i])
            for kk in range(k):
                out[kk, i, :, :c] = torch.randn(p, c, 2)
        lo_of[0] += n
        return out
    ade, fde, per_a, per_f = evaluate_ade_fde_device(m, batches, 20, noise_fn=noise_fn)
    np.testing.assert_allclose(per_a, g["per_ped_ade"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(per_f, g["per_ped_fde"], rtol=0, atol=1e-4)
    assert abs(ade - float(g["ade"])) < 5e-5 and abs(fde - float(g["fde"])) < 5e-5
    # the kernel's own Philox draws: a different sample of the same distribution (181 peds, best of 20)
    ade_p, fde_p, _, _ = evaluate_ade_fde_device(m, batches, 20, seed=1)
    assert abs(ade_p - float(g["ade"])) < 0.15 * float(g["ade"])
    assert abs(fde_p - float(g["fde"])) < 0.20 * float(g["fde"])


@pytest.mark.parametrize("max_norm", (None, 0.05, 1e6))
def test_optim_step_equals_clip_grad_norm_then_sgd(dev, max_norm):
    from social_stgcnn_amd import ops
    g = torch.Generator().manual_seed(2)
    p0 = torch.randn(7563, generator=g)
    g0 = torch.randn(7563, generator=g) * 0.01
    ref_p = torch.nn.Parameter(p0.clone())
    ref_p.grad = g0.clone()
    if max_norm is not None:
        total = torch.nn.utils.clip_grad_norm_([ref_p], max_norm)
    else:
        total = g0.norm()
    with torch.no_grad():
        ref_p -= 0.01 * ref_p.grad
    p, gr = p0.to(dev), g0.to(dev)
    nrm = torch.zeros(1, device=dev)
    lr_dev = torch.full((1,), 0.01, device=dev)
    ops.optim_step(p, gr, lr=123.0, max_norm=max_norm, lr_dev=lr_dev, grad_norm=nrm)      # lr_dev wins
    np.testing.assert_allclose(p.cpu().numpy(), ref_p.detach().numpy(), rtol=0, atol=5e-7)   # 1 ulp: fused multiply-add
    np.testing.assert_allclose(gr.cpu().numpy(), ref_p.grad.numpy(), rtol=1e-5, atol=1e-9)
    assert abs(float(nrm) - float(total)) < 1e-5 * float(total)
    p2 = p0.to(dev)
    ops.optim_step(p2, g0.to(dev), lr=0.01, max_norm=max_norm)                              # host lr
    np.testing.assert_allclose(p2.cpu().numpy(), ref_p.detach().numpy(), rtol=0, atol=5e-7)   # 1 ulp: fused multiply-add


def test_trainer_clip_and_schedule_follow_the_reference_loop(dev):
    """train.py:71-74 + :200: clip_grad_norm_ before the SGD step, StepLR(step, 0.2) once per epoch -- two epochs
    of the fused trainer (second epoch replayed from a captured hipGraph so the device-side lr is exercised)
    against the oracle's group step + torch.nn.utils.clip_grad_norm_ + torch.optim.lr_scheduler.StepLR."""
    import bench
    from oracle import stgcnn_oracle as o
    from social_stgcnn_amd import ops
    from social_stgcnn_amd.model import social_stgcnn
    from social_stgcnn_amd.trainer import Trainer
    n, v, clip, lr0 = 12, 6, 0.02, 0.05
    obs_rel, target = bench.synth_scenes(n, v, 11)
    torch.manual_seed(4)
    m = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12)
    state = o.clone_state({k: t.detach().clone() for k, t in m.state_dict().items()})
    m.to(dev).train()
    nodes, adj = ops.adj_build(torch.from_numpy(obs_rel).to(dev))
    x, tgt = nodes.permute(0, 3, 1, 2), torch.from_numpy(target).to(dev)
    w = torch.full((n,), 1.0 / n, device=dev)
    tr = Trainer(m, lr=lr0, clip_grad=clip, lr_sh_rate=1)
    tr.step(x, adj, tgt, None, w)
    assert abs(tr.scheduler_step() - lr0 * 0.2) < 1e-12
    replay = tr.capture(x, adj, tgt, None, w)
    replay()
    assert abs(tr.scheduler_step() - lr0 * 0.04) < 1e-12
    replay()
    # oracle: every scene in the loss with weight 1/n == reference group of n+1 scenes with batch_size n
    keys = [k for k, _ in m.named_parameters()]
    params = [torch.nn.Parameter(state[k].clone()) for k in keys]
    opt = torch.optim.SGD(params, lr=lr0)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.2)
    xs, As = nodes.cpu(), adj.cpu()
    for _ in range(3):
        for k, prm in zip(keys, params):
            state[k] = prm
        opt.zero_grad()
        loss = 0
        for i in range(n):
            loss = loss + o.scene_loss(state, xs[i].permute(2, 0, 1)[None], As[i], torch.from_numpy(target[i]), True)[0]
        (loss / n).backward()
        torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], clip)
        opt.step()
        sched.step()
    got = m.state_dict()
    for k, prm in zip(keys, params):
        if prm.grad is None:
            continue
        a, b = got[k].cpu().numpy(), prm.detach().numpy()
        assert np.abs(a - b).max() < 2e-5 * max(1.0, np.abs(b).max()), k


@pytest.mark.parametrize("clip", (0.01, None))
def test_step_tail_launch_equals_its_three_separate_launches(dev, clip):
    """stg_train_tail (BatchNorm fold + reported loss + clip/SGD in ONE launch, what Trainer.step runs on one rank with
    clipping) and stg_model_bwd_step (no clipping: the tail rides in the backward's reduction launch)
    against the module-API sequence: forward with its own stg_bn_fold, stg_weighted_sum, stg_optim_step -- same
    inputs, ragged scenes with an empty one; running statistics and parameters within 1 ulp of the summation order."""
    import bench
    from social_stgcnn_amd import ops
    from social_stgcnn_amd.model import social_stgcnn
    from social_stgcnn_amd.trainer import Trainer
    n, v = 37, 9
    obs_rel, target = bench.synth_scenes(n, v, 21)
    nodes, adj = ops.adj_build(torch.from_numpy(obs_rel).to(dev))
    x, tgt = nodes.permute(0, 3, 1, 2), torch.from_numpy(target).to(dev)
    peds = torch.randint(1, v + 1, (n,), generator=torch.Generator().manual_seed(0)).to(torch.int32)
    peds[5] = 0
    peds = peds.to(dev)
    w = torch.rand(n, generator=torch.Generator().manual_seed(1)).to(dev)
    out = []
    for fused in (True, False):
        torch.manual_seed(9)
        m = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12).to(dev).train()
        tr = Trainer(m, lr=0.05, clip_grad=clip)
        if fused:
            total, _, _ = tr.step(x, adj, tgt, peds, w)
        else:
            total, _, _ = tr.forward_backward(x, adj, tgt, peds, w)
            tr._update(m.flat_parameters(), tr._flat_grad())
        out.append((float(total), {k: t.detach().cpu().clone() for k, t in m.state_dict().items()}))
    assert abs(out[0][0] - out[1][0]) <= 1e-6 * abs(out[1][0])
    for k, ref in out[1][1].items():
        got = out[0][1][k]
        if "num_batches" in k:
            assert torch.equal(got, ref), k
        elif "running" in k:                 # same ordered fold, chunked over 1024 instead of 256 threads
            assert torch.allclose(got, ref, rtol=1e-6, atol=1e-7), k
        else:
            assert float((got - ref).abs().max()) <= 2e-7 * max(1.0, float(ref.abs().max())), k
    assert int(out[0][1]["st_gcns.0.tcn.0.num_batches_tracked"]) == n - 1


def test_fused_loss_backward_equals_loss_kernel_then_backward(dev):
    """stg_model_bwd_nll (the backward's input stage computes dV_pred from V_pred and the target) against
    stg_nll_fwd + stg_model_bwd on the same ragged batch (an empty scene, per-scene weights incl. a zero): per-scene
    losses and every parameter gradient, on the wave-per-scene and the workgroup-per-scene kernels; split-bf16 operands
    report 'unsupported' so the trainer falls back."""
    import bench
    from social_stgcnn_amd import ops
    from social_stgcnn_amd.model import social_stgcnn
    n, v = 29, 11
    obs_rel, target = bench.synth_scenes(n, v, 33)
    nodes, adj = ops.adj_build(torch.from_numpy(obs_rel).to(dev))
    x, tgt = nodes.permute(0, 3, 1, 2), torch.from_numpy(target).to(dev)
    peds = torch.randint(1, v + 1, (n,), generator=torch.Generator().manual_seed(5)).to(torch.int32)
    peds[7] = 0
    peds = peds.to(dev)
    w = torch.rand(n, generator=torch.Generator().manual_seed(6))
    w[3] = 0.0
    w = w.to(dev)
    torch.manual_seed(12)
    m = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12).to(dev).train()
    y, _ = m(x, adj, peds)
    losses = ops.backward_from_target(m, y.detach(), tgt, w)
    assert losses is not None
    fused = {k: (None if p.grad is None else p.grad.detach().cpu().clone()) for k, p in m.named_parameters()}
    for p in m.parameters():
        p.grad = None
    state = {k: t.detach().clone() for k, t in m.state_dict().items()}
    torch.manual_seed(12)
    m2 = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12).to(dev).train()
    y2, _ = m2(x, adj, peds)
    l2, dy = ops.bivariate_nll_with_grad(y2.detach(), tgt, peds, w)
    y2.backward(dy)
    assert torch.allclose(losses.cpu(), l2.cpu(), rtol=2e-6, atol=1e-6)
    assert float(losses[7]) == 0.0
    gmax = max(float(p.grad.abs().max()) for p in m2.parameters() if p.grad is not None)
    for k, p in m2.named_parameters():
        if p.grad is None:
            assert fused[k] is None, k
            continue
        ref = p.grad.detach().cpu()
        # (biases in front of a train-mode BatchNorm have an exactly-zero true gradient: both sides hold rounding noise
        # there, compared on the scale of the largest gradient)
        scale = max(0.05 * gmax, float(ref.abs().max()))
        assert float((fused[k] - ref).abs().max()) <= 2e-6 * scale, k
    del state
    # the workgroup-per-scene kernels (small batches) fuse the loss the same way; split-bf16 operands: nothing fused,
    # the caller is told so
    for waves in (0, 1, 4):
        ops.OPTIONS["wg_path"], ops.OPTIONS["wg_waves"] = True, waves
        try:
            for p in m2.parameters():
                p.grad = None
            y3, _ = m2(x, adj, peds)
            l3 = ops.backward_from_target(m2, y3.detach(), tgt, w)
            assert l3 is not None
            assert torch.allclose(l3.cpu(), l2.cpu(), rtol=2e-6, atol=1e-6)
            assert float(l3[7]) == 0.0
            for k, p in m2.named_parameters():
                if fused[k] is None:
                    assert p.grad is None, k
                    continue
                # (other kernels, another summation order: compared on the scale of the largest gradient -- the
                # rounding noise of an exactly-zero bias gradient is of that scale)
                assert float((p.grad.detach().cpu() - fused[k]).abs().max()) <= 1e-5 * gmax, (waves, k)
        finally:
            ops.OPTIONS["wg_path"], ops.OPTIONS["wg_waves"] = False, 0
    ops.OPTIONS["split_bf16"] = True
    try:
        y4, _ = m2(x, adj, peds)
        assert ops.backward_from_target(m2, y4.detach(), tgt, w) is None
    finally:
        ops.OPTIONS["split_bf16"] = False
