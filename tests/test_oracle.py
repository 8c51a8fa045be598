"""CPU: pin oracle/stgcnn_oracle.py against fixtures produced by the reference itself
(tests/golden/make_golden.py).  Tolerances are fp32 rounding-level (the oracle and the
reference call the same torch CPU ops; the adjacency closed form differs from networkx's
sparse product only in summation order)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import stgcnn_oracle as O

torch.set_num_threads(1)
VS = (2, 3, 5, 8, 17, 32, 57)


def _state(npz, prefix=""):
    return {k[len(prefix):]: torch.from_numpy(np.array(npz[k])) for k in npz.files
            if k.startswith(prefix)}


@pytest.mark.parametrize("v", VS + ("tie",))
def test_adjacency_vectorised(v):
    g = load_golden("adj_cases.npz")
    nodes, lap = O.seq_to_graph_np(g["rel_%s" % v])
    assert np.array_equal(nodes, g["nodes_%s" % v])
    np.testing.assert_allclose(lap, g["lap_%s" % v], rtol=0, atol=2e-7)
    assert np.all(lap[0] == 0)            # rel[0] == 0 -> A = I -> L = 0 (SURVEY 3.4 ii)


@pytest.mark.parametrize("v", (2, 3, 5, 8, "tie"))
def test_adjacency_loops(v):
    g = load_golden("adj_cases.npz")
    nodes, lap = O.seq_to_graph_loops(g["rel_%s" % v])
    assert np.array_equal(nodes.numpy(), g["nodes_%s" % v])
    np.testing.assert_allclose(lap.numpy(), g["lap_%s" % v], rtol=0, atol=2e-7)


def test_adjacency_pred_window():
    g = load_golden("adj_cases.npz")
    for v in (5, 17):
        nodes, lap = O.seq_to_graph_np(g["predrel_%d" % v])
        assert np.array_equal(nodes, g["prednodes_%d" % v])
        np.testing.assert_allclose(lap, g["predlap_%d" % v], rtol=0, atol=2e-7)


def test_tie_case_has_zero_edges():
    g = load_golden("adj_cases.npz")
    lap = g["lap_tie"]
    assert lap[2, 0, 1] == 0 and lap[2, 1, 0] == 0       # equal velocities -> no edge


@pytest.mark.parametrize("v", VS)
def test_eval_forward(v):
    w = _state(load_golden("weights_eth.npz"))
    a = load_golden("adj_cases.npz")
    f = load_golden("forward_eval.npz")
    x = torch.from_numpy(a["nodes_%d" % v]).unsqueeze(0).permute(0, 3, 1, 2)
    A = torch.from_numpy(a["lap_%d" % v])
    with torch.no_grad():
        g = O.conv_temporal_graphical(w, "st_gcns.0.gcn", x, A)
        h = O.st_gcn_forward(w, "st_gcns.0", x, A, False)
        y = O.social_stgcnn_forward(w, x, A, False)
    np.testing.assert_allclose(g.numpy(), f["gcn_%d" % v], rtol=0, atol=1e-6)
    np.testing.assert_allclose(h.numpy(), f["stgcn_%d" % v], rtol=0, atol=2e-6)
    np.testing.assert_allclose(y.numpy(), f["vpred_%d" % v], rtol=0, atol=5e-6)


@pytest.mark.parametrize("v", (3, 17, 57))
def test_train_forward_backward(v):
    t = load_golden("train_fwd_bwd.npz")
    a = load_golden("adj_cases.npz")
    state = _state(t, "sd_%d/" % v)
    keys = [k for k in state if "running" not in k and "num_batches" not in k]
    params = {k: state[k].clone().requires_grad_(True) for k in keys}
    work = dict(state)
    work.update(params)
    x = torch.from_numpy(a["nodes_%d" % v]).unsqueeze(0).permute(0, 3, 1, 2)
    A = torch.from_numpy(a["lap_%d" % v])
    tgt = torch.from_numpy(a["prednodes_%d" % v])
    loss, v_pred = O.scene_loss(work, x, A, tgt, True)
    loss.backward()
    np.testing.assert_allclose(v_pred.detach().permute(2, 0, 1).unsqueeze(0).numpy(),
                               t["vpred_%d" % v], rtol=0, atol=5e-6)
    assert abs(loss.item() - float(t["loss_%d" % v])) < 1e-5
    for k in keys:
        ref = t["grad_%d/%s" % (v, k)]
        if np.isnan(ref).all():
            assert params[k].grad is None, k          # dead parameters
            continue
        got = params[k].grad.numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-6, err_msg=k)
    for k in state:
        if "running" in k:
            np.testing.assert_allclose(work[k].numpy(), t["after_%d/%s" % (v, k)],
                                       rtol=1e-6, atol=1e-7, err_msg=k)
        if "num_batches" in k:
            assert int(work[k]) == int(t["after_%d/%s" % (v, k)])


def test_bivariate_loss_and_grad():
    g = load_golden("loss_cases.npz")
    vp = torch.from_numpy(g["vpred"]).requires_grad_(True)
    loss = O.bivariate_loss(vp, torch.from_numpy(g["vtrgt"]))
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    np.testing.assert_allclose(vp.grad.numpy(), g["dvpred"], rtol=1e-5, atol=1e-7)
    assert np.all(g["dvpred"][0, 0] == 0)            # clamp active -> zero gradient


def test_state_dict_keys_match_reference():
    w = load_golden("weights_eth.npz")
    assert list(w.files) == O.state_dict_keys(1, 5)
    assert len(w.files) == 40


def test_group_boundaries():
    assert O.group_boundaries(40, 16) == [15, 31, 39]
    assert O.group_boundaries(2785, 128)[-1] == 2784
    assert O.group_boundaries(256, 128) == [127, 255]


def test_train_loop_matches_reference_train():
    """R9: the reference's own train() (train.py:28-79) on 40 eth/test scenes, batch_size 16."""
    g = load_golden("train_loop.npz")
    e = load_golden("eth_test_windows.npz")
    state = _state(g, "before/")
    keys = [k for k in state if "running" not in k and "num_batches" not in k]
    n_sc, bs, lr = int(g["n_scenes"]), int(g["batch_size"]), float(g["lr"])
    starts = np.concatenate([[0], np.cumsum(e["num_peds"])])
    scenes = []
    for i in range(n_sc):
        rel = e["seq_rel"][starts[i]:starts[i + 1]]
        nodes, lap = O.seq_to_graph_np(rel[:, :, :8])
        tgt, _ = O.seq_to_graph_np(rel[:, :, 8:])
        scenes.append((torch.from_numpy(nodes).unsqueeze(0).permute(0, 3, 1, 2),
                       torch.from_numpy(lap), torch.from_numpy(tgt)))
    bounds = O.group_boundaries(n_sc, bs)
    total, lo = 0.0, 0
    for b in bounds:
        loss, _ = O.train_group(state, scenes[lo:b + 1], bs, lr, keys)
        total += loss
        lo = b + 1
    assert abs(total / n_sc - float(g["epoch_loss"])) < 1e-6
    for k in state:
        ref = g["after/" + k]
        if "num_batches" in k:
            assert int(state[k]) == int(ref)
        else:
            np.testing.assert_allclose(state[k].detach().numpy(), ref, rtol=2e-5, atol=2e-6,
                                       err_msg=k)


def test_eval_ade_fde_matches_reference_test():
    """R10: reference test() (test.py:18-127) on eth/test, torch.manual_seed(0), 20 samples."""
    g = load_golden("eval_ade_fde.npz")
    e = load_golden("eth_test_windows.npz")
    starts = np.concatenate([[0], np.cumsum(e["num_peds"])])
    vp = torch.from_numpy(g["vpred_cat"])
    torch.manual_seed(0)
    ades, fdes = [], []
    for i in range(len(e["num_peds"])):
        s, t = starts[i], starts[i + 1]
        seq, rel = e["seq"][s:t], e["seq_rel"][s:t]
        a, f = O.best_of_k_errors(vp[:, s:t], seq[:, :, 7], np.transpose(rel[:, :, 8:], (2, 0, 1)),
                                  20)
        ades += a
        fdes += f
    np.testing.assert_allclose(ades, g["per_ped_ade"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(fdes, g["per_ped_fde"], rtol=0, atol=2e-5)
    assert abs(np.mean(ades) - float(g["ade"])) < 1e-5
    assert abs(np.mean(fdes) - float(g["fde"])) < 1e-5


def test_best_of_k_noise_variant_is_the_sampler_variant():
    """oracle.best_of_k_errors_noise (the checker of the device evaluation op) == oracle.best_of_k_errors (pinned
    by the reference's test() golden) when eps is the CPU generator's own stream."""
    from oracle import stgcnn_oracle as o
    g = np.random.default_rng(0)
    p, v, k = 12, 5, 20
    torch.manual_seed(3)
    vp = torch.randn(p, v, 5) * 0.5
    obs = (g.normal(size=(v, 2)) * 5).astype(np.float32)
    tgt = (g.normal(size=(p, v, 2)) * 0.3).astype(np.float32)
    torch.manual_seed(11)
    a1, f1 = o.best_of_k_errors(vp, obs, tgt, k)
    torch.manual_seed(11)
    eps = torch.stack([torch.randn(p, v, 2) for _ in range(k)])
    a2, f2 = o.best_of_k_errors_noise(vp, obs, tgt, eps)
    assert a1 == a2 and f1 == f2


def test_oracle_reproduces_the_reference_training_curve():
    """train_curve.npz (reference train()/vald(), 6 epochs, clip 0.5, StepLR(2, 0.2)): the oracle's group step +
    torch's own clip_grad_norm_ / StepLR walk the same curve -- first two epochs here (CPU time)."""
    from oracle import stgcnn_oracle as o
    g = load_golden("train_curve.npz")
    e = load_golden("eth_test_windows.npz")
    n_sc, bs = int(g["n_scenes"]), int(g["batch_size"])
    starts = np.concatenate([[0], np.cumsum(e["num_peds"])])
    scenes = []
    for i in range(n_sc):
        rel = e["seq_rel"][starts[i]:starts[i + 1]]                    # (V,2,20)
        nodes, lap = o.seq_to_graph_np(rel[:, :, :8])
        tgt = np.ascontiguousarray(np.transpose(rel[:, :, 8:], (2, 0, 1)))
        scenes.append((torch.from_numpy(nodes).unsqueeze(0).permute(0, 3, 1, 2), torch.from_numpy(lap),
                       torch.from_numpy(tgt)))
    state = {k[len("before/"):]: torch.from_numpy(np.array(g[k])) for k in g.files if k.startswith("before/")}
    keys = [k for k in state if not any(s in k for s in ("running", "num_batches"))]
    params = [torch.nn.Parameter(state[k].clone()) for k in keys]
    for k, p in zip(keys, params):
        state[k] = p
    opt = torch.optim.SGD(params, lr=float(g["lr"]))
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=int(g["lr_sh_rate"]), gamma=0.2)
    bounds = o.group_boundaries(n_sc, bs)
    for ep in range(2):
        loss_sum, lo = 0.0, 0
        for b in bounds:
            opt.zero_grad()
            tot = 0
            for i in range(lo, b + 1):
                l, _ = o.scene_loss(state, *scenes[i], True)          # closing scene: forward only (BN stats)
                if i != b:
                    tot = tot + l
            # train.py:58-74: the scene that closes a group is forwarded but left out of the loss
            loss = tot / bs
            loss.backward()
            torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], float(g["clip_grad"]))
            opt.step()
            loss_sum += float(loss.detach())
            lo = b + 1
        assert abs(loss_sum / n_sc - float(g["train_loss"][ep])) < 2e-6, (ep, loss_sum / n_sc)
        sched.step()
