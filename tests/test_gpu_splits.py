"""GPU (-m gpu): all five ETH/UCY splits and the real eth/train batch against the reference's own outputs
(fixtures of tests/golden/make_golden_splits.py: reference test() with torch.manual_seed(0) per split, reference
train() for one eth/train epoch at batch_size 512), and the reference training loop replayed through the plain
nn.Module API + torch.optim.SGD (train.py:36-77)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu
SPLITS = ("eth", "hotel", "univ", "zara1", "zara2")
CFG = dict(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12)


def _state(npz, prefix=""):
    return {k[len(prefix):]: torch.from_numpy(np.array(npz[k])) for k in npz.files if k.startswith(prefix)}


def _windows(name, files=None):
    from social_stgcnn_amd import data
    return data.load_windows(os.path.join(GOLDEN, "data", name), 8, 12, 1, with_non_linear=False, files=files)


def _device_batch(dev, win, idx):
    """windows idx -> (x (N,2,8,V) strided, adj, num_peds, obs_last (N,V,2), target_rel (N,12,V,2)); the graphs are
    built by the adj_build kernel from the (N,8,V,2) collation viewed as (N,V,2,8)."""
    from social_stgcnn_amd import data, ops
    obs_rel, pred_rel, obs_abs, _, counts = data.pad_batch(win, idx)
    peds = torch.from_numpy(counts).to(dev)
    rel_d = torch.from_numpy(obs_rel).to(dev).permute(0, 2, 3, 1)          # (N,V,2,T) view of (N,T,V,2)
    nodes, adj = ops.adj_build(rel_d, peds)
    return nodes.permute(0, 3, 1, 2), adj, peds, obs_abs[:, -1], pred_rel


def _model(dev, name):
    from social_stgcnn_amd.model import social_stgcnn
    m = social_stgcnn(**CFG)
    m.load_state_dict(_state(load_golden("weights_%s.npz" % name)))
    return m.to(dev).eval()


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def splits(dev):
    """per split: (model, windows, list of device batches of <= 64 windows, fixture dict)"""
    g = load_golden("eval_splits.npz")
    out = {}
    for name in SPLITS:
        win = _windows(name + "_test")
        want = g[name + "/num_peds"]
        if not np.array_equal(win.num_peds, want):      # the reference walked the split's files in another order
            files = sorted(os.listdir(os.path.join(GOLDEN, "data", name + "_test")))[::-1]
            win = _windows(name + "_test", files)
        assert np.array_equal(win.num_peds, want), name
        batches = [_device_batch(dev, win, np.arange(lo, min(len(win), lo + 64))) for lo in range(0, len(win), 64)]
        out[name] = (_model(dev, name), win, batches, {k[len(name) + 1:]: g[k] for k in g.files if k.startswith(name + "/")})
    return out


@pytest.mark.parametrize("name", SPLITS)
def test_vpred_of_every_split_matches_the_reference(splits, name):
    """R5 on real ragged batches (univ: up to 57 pedestrians): the five Gaussian parameters of every k-th window
    vs the reference's CPU forward with the shipped checkpoint of that split; north-star bar 1e-4."""
    m, win, batches, g = splits[name]
    every, ref = int(g["vpred_every"]), g["vpred_cat"]
    got = []
    with torch.no_grad():
        for b, (x, adj, peds, _, _) in enumerate(batches):
            y, _ = m(x, adj, peds)
            y = y.permute(0, 2, 3, 1).cpu().numpy()                        # (N,P,V,5)
            for j in range(y.shape[0]):
                if (b * 64 + j) % every == 0:
                    got.append(y[j, :, :int(win.num_peds[b * 64 + j])])
    got = np.concatenate(got, axis=1)
    assert got.shape == ref.shape
    err = float(np.abs(got - ref).max())
    print("%s: max |V_pred - reference| = %.2e over %d pedestrians" % (name, err, ref.shape[1]))
    assert err < 1e-4, err


@pytest.mark.parametrize("name", SPLITS)
def test_cpu_sampler_evaluation_reproduces_reference_ade_fde(splits, name):
    """R10 per split: evaluate_ade_fde (GPU forward, the reference's CPU sampler in the reference's order) with
    torch.manual_seed(0) -> the reference test()'s per-pedestrian best-of-20 ADE / FDE."""
    from social_stgcnn_amd.trainer import evaluate_ade_fde
    m, _, batches, g = splits[name]
    torch.manual_seed(0)
    ade, fde, per_a, per_f = evaluate_ade_fde(m, batches, 20)
    np.testing.assert_allclose(per_a, g["per_ped_ade"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(per_f, g["per_ped_fde"], rtol=0, atol=1e-4)
    assert abs(ade - float(g["ade"])) < 2e-5 and abs(fde - float(g["fde"])) < 2e-5


@pytest.mark.parametrize("name", SPLITS)
def test_device_evaluation_reproduces_reference_ade_fde(splits, name):
    """R10 per split through stg_bestofk_eval, fed the reference sampler's own standard-normal stream
    (torch.manual_seed(0); per scene and sample one (P, V_i, 2) draw, test.py:87-89)."""
    from social_stgcnn_amd.trainer import evaluate_ade_fde_device
    m, win, batches, g = splits[name]
    torch.manual_seed(0)
    at = [0]

    def noise_fn(b, shape):
        k, n, p, v, _ = shape
        out = torch.zeros(shape)
        for i in range(n):
            c = int(win.num_peds[at[0] + i])
            for kk in range(k):
                out[kk, i, :, :c] = torch.randn(p, c, 2)
        at[0] += n
        return out
    ade, fde, per_a, per_f = evaluate_ade_fde_device(m, batches, 20, noise_fn=noise_fn)
    np.testing.assert_allclose(per_a, g["per_ped_ade"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(per_f, g["per_ped_fde"], rtol=0, atol=2e-4)
    assert abs(ade - float(g["ade"])) < 5e-5 and abs(fde - float(g["fde"])) < 5e-5


def test_eth_train_epoch_at_batch_512_equals_the_reference(dev):
    """BASELINE configs[1] on the real data: one reference train() epoch over the 2,785 eth/train windows in the
    reference's dataset order, batch_size 512, SGD lr 0.01 -- six optimizer steps over ragged groups (2..57
    pedestrians per window)."""
    from social_stgcnn_amd.model import social_stgcnn
    from social_stgcnn_amd.trainer import Trainer
    g = load_golden("eth_train_epoch.npz")
    win = _windows("eth_train", [str(f) for f in g["listdir_order"]])
    assert len(win) == int(g["n_scenes"]) == 2785 and np.array_equal(win.num_peds, g["num_peds"])
    m = social_stgcnn(**CFG)
    m.load_state_dict(_state(g, "before/"))
    m.to(dev)

    def batcher(lo, hi):
        x, adj, peds, _, tgt = _device_batch(dev, win, np.arange(lo, hi))
        return x, adj, torch.from_numpy(tgt).to(dev), peds
    tr = Trainer(m, lr=float(g["lr"]))
    ep_loss = tr.train_epoch(batcher, len(win), int(g["batch_size"]))
    print("eth/train epoch loss %.9f (reference %.9f)" % (ep_loss, float(g["epoch_loss"])))
    assert abs(ep_loss - float(g["epoch_loss"])) < 2e-6 * max(1.0, abs(float(g["epoch_loss"])) * 1e3)
    before, bad = _state(g, "before/"), {}
    for k, val in m.state_dict().items():
        ref = g["after/" + k]
        if "num_batches" in k:
            assert int(val) == int(ref), k
            continue
        upd_ref = ref - before[k].numpy()
        upd = val.cpu().numpy() - before[k].numpy()
        scale = max(float(np.abs(upd_ref).max()), 1e-7)
        err = float(np.abs(upd - upd_ref).max())
        if err > 1e-3 * scale + 2e-7:
            bad[k] = (err, scale)
    assert not bad, bad


def test_reference_train_loop_through_the_module_api(dev):
    """train.py:36-77 verbatim in structure, through the drop-in nn.Module surface: N = 1 forwards, `loss += l` over a
    group, ONE backward, clip-free torch.optim.SGD.step() on parameters that are views of the flat buffer,
    optimizer.zero_grad() every iteration -- against the reference's own train() result (train_loop.npz)."""
    from social_stgcnn_amd.metrics import bivariate_loss as graph_loss
    from social_stgcnn_amd.model import social_stgcnn
    from social_stgcnn_amd.utils import seq_to_graph
    g = load_golden("train_loop.npz")
    e = load_golden("eth_test_windows.npz")
    n_sc, batch_size = int(g["n_scenes"]), int(g["batch_size"])
    model = social_stgcnn(**CFG)
    model.load_state_dict(_state(g, "before/"))
    model.to(dev)
    optimizer = torch.optim.SGD(model.parameters(), lr=float(g["lr"]))
    starts = np.concatenate([[0], np.cumsum(e["num_peds"])])
    loader = []
    for i in range(n_sc):
        rel = torch.from_numpy(e["seq_rel"][starts[i]:starts[i + 1]]).to(dev)          # (V,2,20)
        v_obs, a_obs = seq_to_graph(None, rel[:, :, :8], True)
        v_tr, a_tr = seq_to_graph(None, rel[:, :, 8:], True)
        loader.append((v_obs.unsqueeze(0), a_obs.unsqueeze(0), v_tr.unsqueeze(0), a_tr.unsqueeze(0)))
    model.train()
    loss_batch, batch_count, is_fst_loss = 0, 0, True
    turn_point = int(n_sc / batch_size) * batch_size + n_sc % batch_size - 1
    for cnt, (V_obs, A_obs, V_tr, A_tr) in enumerate(loader):
        batch_count += 1
        optimizer.zero_grad()
        V_obs_tmp = V_obs.permute(0, 3, 1, 2)
        V_pred, _ = model(V_obs_tmp, A_obs.squeeze())
        V_pred = V_pred.permute(0, 2, 3, 1)
        V_tr = V_tr.squeeze()
        V_pred = V_pred.squeeze()
        if batch_count % batch_size != 0 and cnt != turn_point:
            l = graph_loss(V_pred, V_tr)
            if is_fst_loss:
                loss = l
                is_fst_loss = False
            else:
                loss += l
        else:
            loss = loss / batch_size
            is_fst_loss = True
            loss.backward()
            optimizer.step()
            loss_batch += loss.item()
    ep_loss = loss_batch / batch_count
    assert abs(ep_loss - float(g["epoch_loss"])) < 2e-6, (ep_loss, float(g["epoch_loss"]))
    before, bad = _state(g, "before/"), {}
    for k, val in model.state_dict().items():
        ref = g["after/" + k]
        if "num_batches" in k:
            assert int(val) == int(ref), k
            continue
        upd_ref = ref - before[k].numpy()
        upd = val.cpu().numpy() - before[k].numpy()
        scale = max(float(np.abs(upd_ref).max()), 1e-7)
        err = float(np.abs(upd - upd_ref).max())
        if err > 5e-4 * scale + 2e-7:
            bad[k] = (err, scale)
    assert not bad, bad
    # the three dead parameters keep grad None like the reference (model.py:191)
    assert model.tpcnns[4].weight.grad is None and model.prelus[4].weight.grad is None


def test_checkpoint_round_trip_into_the_reference(dev, tmp_path):
    """N4 on the GPU.  (a) The weights of the fixture checkpoint (written by trainer.Checkpoint, READ BY THE REFERENCE via
    test.py:153-186 in tests/golden/make_golden_ckpt.py) on the device -> Checkpoint.record -> load_checkpoint into a
    fresh model: V_pred on eth/test equals the in-memory model bit for bit and the reference's V_pred of that file to
    2e-5, its ADE / FDE with the reference's seed-0 draws.  (b) Two training groups on the GPU -> record -> reload:
    the reloaded model is the trained one, bit for bit (train.py:228-246 writer, test.py:158 reader)."""
    import argparse
    from social_stgcnn_amd.model import social_stgcnn
    from social_stgcnn_amd.trainer import Checkpoint, Trainer, evaluate_ade_fde, load_checkpoint
    g = load_golden("ckpt_roundtrip.npz")
    win = _windows("eth_test")
    assert np.array_equal(win.num_peds, g["num_peds"])
    batches = [_device_batch(dev, win, np.arange(lo, min(len(win), lo + 64))) for lo in range(0, len(win), 64)]

    def vpred(model):
        model.eval()
        got = []
        with torch.no_grad():
            for b, (x, adj, peds, _, _) in enumerate(batches):
                y, _ = model(x, adj, peds)
                y = y.permute(0, 2, 3, 1).cpu().numpy()
                got += [y[j, :, :int(win.num_peds[b * 64 + j])] for j in range(y.shape[0])]
        return np.concatenate(got, axis=1)

    m = social_stgcnn(**CFG)
    m.load_state_dict(_state(g, "sd/"))
    m.to(dev)
    d = str(tmp_path / "checkpoint" / "social-stgcnn-roundtrip")
    assert Checkpoint(d, argparse.Namespace(dataset="eth", n_stgcnn=1, n_txpcnn=5)).record(0, m, 0.5, 0.25)
    m2 = load_checkpoint(social_stgcnn(**CFG), d + "/val_best.pth").to(dev)
    v1, v2 = vpred(m), vpred(m2)
    assert np.array_equal(v1, v2)
    err = float(np.abs(v2 - g["vpred_cat"]).max())
    print("checkpoint round trip: max |V_pred - reference reading the same file| = %.2e" % err)
    assert v2.shape == g["vpred_cat"].shape and err < 2e-5, err
    torch.manual_seed(0)
    ade, fde, per_a, per_f = evaluate_ade_fde(m2, batches, 20)
    np.testing.assert_allclose(per_a, g["per_ped_ade"], rtol=0, atol=1e-4)
    assert abs(ade - float(g["ade"])) < 2e-5 and abs(fde - float(g["fde"])) < 2e-5
    # (b) train two groups, save, reload
    m.train()
    tr = Trainer(m, lr=0.01)
    x, adj, peds, _, tgt = batches[0]
    tgt = torch.from_numpy(tgt).to(dev)
    w = torch.full((x.shape[0],), 1.0 / x.shape[0], device=dev)
    for _ in range(2):
        tr.step(x, adj, tgt, peds, w)
    ck = Checkpoint(str(tmp_path / "checkpoint" / "trained"), argparse.Namespace(dataset="eth"))
    assert ck.record(0, m, 0.1, 0.2) and not ck.record(1, m, 0.1, 0.3)
    m3 = load_checkpoint(social_stgcnn(**CFG), ck.dir + "/val_best.pth").to(dev)
    for (k, a), (_, b3) in zip(m.state_dict().items(), m3.state_dict().items()):
        assert torch.equal(a, b3), k
    assert int(m3.state_dict()["st_gcns.0.tcn.0.num_batches_tracked"]) == 1234 + 2 * x.shape[0]
    assert np.array_equal(vpred(m), vpred(m3))
