"""GPU (-m gpu): N1 on the device -- the windowed dataset resident in HBM, batches gathered by device index
(stg_gather_windows), and a whole reference-style epoch replayed from ONE captured hipGraph (gather -> adj_build ->
forward -> loss -> backward -> update) with no host->device traffic in the loop."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu
CFG = dict(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12)


def _state(npz, prefix=""):
    return {k[len(prefix):]: torch.from_numpy(np.array(npz[k])) for k in npz.files if k.startswith(prefix)}


@pytest.fixture(scope="module")
def eth_train():
    from social_stgcnn_amd import data
    g = load_golden("eth_train_epoch.npz")
    win = data.load_windows(os.path.join(GOLDEN, "data", "eth_train"), 8, 12, 1, with_non_linear=False,
                            files=[str(f) for f in g["listdir_order"]])
    return g, win


def test_device_gather_is_bit_equal_to_the_host_collation(eth_train):
    from social_stgcnn_amd import data
    from social_stgcnn_amd.dataset import DeviceWindows
    _, win = eth_train
    dev = torch.device("cuda", 0)
    ds = DeviceWindows(win, dev)
    assert len(ds) == 2785 and ds.v_max == 57
    rng = np.random.default_rng(0)
    for n in (1, 7, 300):
        idx = rng.choice(len(win), size=n, replace=False).astype(np.int32)
        obs_rel, target, peds = ds.gather(torch.from_numpy(idx).to(dev))
        h_obs, h_pred, _, _, counts = data.pad_batch(win, idx, v_pad=57)          # (N,8,V,2), (N,12,V,2)
        assert np.array_equal(peds.cpu().numpy(), counts)
        assert np.array_equal(obs_rel.cpu().numpy(), np.transpose(h_obs, (0, 2, 3, 1)))
        assert np.array_equal(target.cpu().numpy(), h_pred)
    # default index = the first n windows; a smaller pad truncates nothing it should not
    obs_rel, _, peds = ds.gather(n=5)
    assert np.array_equal(peds.cpu().numpy(), win.num_peds[:5].astype(np.int32))


def test_captured_device_epoch_equals_the_reference_epoch(eth_train):
    """The reference's train() epoch over eth/train in dataset order (batch_size 512, fixture eth_train_epoch.npz)
    through EpochRunner: five replays of the captured 512-scene group + one of the captured tail group, indices refreshed
    on the device."""
    from social_stgcnn_amd.dataset import DeviceWindows, EpochRunner
    from social_stgcnn_amd.model import social_stgcnn
    from social_stgcnn_amd.trainer import Trainer
    g, win = eth_train
    dev = torch.device("cuda", 0)
    m = social_stgcnn(**CFG)
    m.load_state_dict(_state(g, "before/"))
    m.to(dev)
    ds = DeviceWindows(win, dev)
    runner = EpochRunner(Trainer(m, lr=float(g["lr"])), ds, int(g["batch_size"]))
    order = torch.arange(len(ds), device=dev, dtype=torch.int32)
    ep_loss = float(runner.train_epoch(order))
    assert abs(ep_loss - float(g["epoch_loss"])) < 2e-6, (ep_loss, float(g["epoch_loss"]))
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
    # a shuffled second and third epoch run from the same captured graphs (one per group size) and must equal the same
    # epochs stepped eagerly, group by group (everything a graph reads -- indices, loss weights -- is still what it was
    # captured with)
    from social_stgcnn_amd import ops
    from social_stgcnn_amd.trainer import group_bounds, group_weights
    bs = int(g["batch_size"])
    m2 = social_stgcnn(**CFG)
    m2.load_state_dict({k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    m2.to(dev).train()
    tr2 = Trainer(m2, lr=float(g["lr"]))
    gen = torch.Generator(device=dev).manual_seed(1)
    for _ in range(2):
        order2 = ds.shuffled_order(gen)
        assert sorted(order2.cpu().tolist()) == list(range(len(ds)))
        ep_graph = float(runner.train_epoch(order2))
        lo, tot = 0, 0.0
        for b in group_bounds(len(ds), bs):
            idx = order2[lo:b + 1].contiguous()
            obs_rel, target, peds = ds.gather(idx, v_pad=runner.obs_rel.shape[1])
            nodes, adj = ops.adj_build(obs_rel, peds)
            t, _, _ = tr2.step(nodes.permute(0, 3, 1, 2), adj, target, peds, group_weights(b + 1 - lo, bs, dev))
            tot += float(t)
            lo = b + 1
        assert abs(ep_graph - tot / len(ds)) < 1e-6, (ep_graph, tot / len(ds))
    for (k, a), (_, b2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert float((a.float() - b2.float()).abs().max()) < 1e-5, k
