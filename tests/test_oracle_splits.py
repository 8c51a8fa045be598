"""CPU: the oracle and the host ingest against the five-split / eth-train fixtures the reference produced
(tests/golden/make_golden_splits.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from oracle import stgcnn_oracle as O
from social_stgcnn_amd import data

SPLITS = ("eth", "hotel", "univ", "zara1", "zara2")
# windows / pedestrians / largest crowd per test split and for eth/train, measured with the reference's
# TrajectoryDataset
COUNTS = {"eth_test": (70, 181, 5), "hotel_test": (301, 1053, 8), "univ_test": (947, 24334, 57),
          "zara1_test": (602, 2253, 14), "zara2_test": (921, 5833, 14), "eth_train": (2785, 29809, 57)}


def _state(npz, prefix=""):
    return {k[len(prefix):]: torch.from_numpy(np.array(npz[k])) for k in npz.files if k.startswith(prefix)}


def _windows(name, files=None):
    return data.load_windows(os.path.join(GOLDEN, "data", name), 8, 12, 1, with_non_linear=False, files=files)


@pytest.mark.parametrize("name", sorted(COUNTS))
def test_ingest_window_counts(name):
    w = _windows(name)
    assert (len(w), int(w.num_peds.sum()), int(w.num_peds.max())) == COUNTS[name]


def test_eth_train_windows_follow_the_reference_file_order():
    g = load_golden("eth_train_epoch.npz")
    w = _windows("eth_train", [str(f) for f in g["listdir_order"]])
    assert np.array_equal(w.num_peds, g["num_peds"])


@pytest.mark.parametrize("name", SPLITS)
def test_oracle_vpred_on_every_split(name):
    """oracle forward (eval mode, the split's shipped weights) on every k-th test window == the reference's V_pred."""
    g = load_golden("eval_splits.npz")
    w = _windows(name + "_test")
    assert np.array_equal(w.num_peds, g[name + "/num_peds"])
    state = _state(load_golden("weights_%s.npz" % name))
    every, ref = int(g[name + "/vpred_every"]), g[name + "/vpred_cat"]
    idx = list(range(0, len(w), every))
    if name == "univ":
        idx = idx[::4]                       # CPU time: 15 of the 60 fixture windows (V up to 57)
    col = 0
    cols = {}
    for i in range(0, len(w), every):
        cols[i] = col
        col += int(w.num_peds[i])
    worst = 0.0
    with torch.no_grad():
        for i in idx:
            s, e = w.seq_start_end[i]
            nodes, lap = O.seq_to_graph_np(w.seq_rel[s:e, :, :8].astype(np.float32))
            y = O.social_stgcnn_forward(state, torch.from_numpy(nodes).unsqueeze(0).permute(0, 3, 1, 2),
                                        torch.from_numpy(lap), False)
            vp = y[0].permute(1, 2, 0).numpy()                      # (P,V,5)
            worst = max(worst, float(np.abs(vp - ref[:, cols[i]:cols[i] + (e - s)]).max()))
    assert worst < 2e-5, worst


def test_oracle_reproduces_the_reference_eth_train_epoch():
    """BASELINE configs[1] data: the reference's train() over the 2,785 eth/train windows (dataset order, batch_size
    512, six optimizer steps) through the oracle's group step: epoch loss and every weight / running statistic."""
    g = load_golden("eth_train_epoch.npz")
    w = _windows("eth_train", [str(f) for f in g["listdir_order"]])
    state = _state(g, "before/")
    keys = [k for k in state if "running" not in k and "num_batches" not in k]
    bs = int(g["batch_size"])
    bounds = O.group_boundaries(len(w), bs)
    assert bounds == [511, 1023, 1535, 2047, 2559, 2784]
    torch.set_num_threads(1)
    scenes = []
    for i in range(len(w)):
        s, e = w.seq_start_end[i]
        rel = w.seq_rel[s:e].astype(np.float32)
        nodes, lap = O.seq_to_graph_np(rel[:, :, :8])
        tgt, _ = O.seq_to_graph_np(rel[:, :, 8:])
        scenes.append((torch.from_numpy(nodes).unsqueeze(0).permute(0, 3, 1, 2), torch.from_numpy(lap),
                       torch.from_numpy(tgt)))
    total, lo = 0.0, 0
    for b in bounds:
        loss, _ = O.train_group(state, scenes[lo:b + 1], bs, float(g["lr"]), keys)
        total += loss
        lo = b + 1
    assert abs(total / len(w) - float(g["epoch_loss"])) < 1e-7
    for k in state:
        ref = g["after/" + k]
        if "num_batches" in k:
            assert int(state[k]) == int(ref)
        else:
            np.testing.assert_allclose(state[k].detach().numpy(), ref, rtol=2e-5, atol=2e-6, err_msg=k)


# windows / pedestrians / largest crowd of all 15 split directories, measured with the reference's TrajectoryDataset
# (SURVEY 8d): the known answers of the ingest.  The full dataset only exists next to the reference (build container).
ALL_COUNTS = {"eth": ((2785, 29809, 57), (660, 5349, 42), (70, 181, 5)),
              "hotel": ((2594, 29152, 57), (621, 5136, 42), (301, 1053, 8)),
              "univ": ((2076, 9231, 14), (530, 2708, 13), (947, 24334, 57)),
              "zara1": ((2322, 28010, 57), (605, 5118, 42), (602, 2253, 14)),
              "zara2": ((2112, 25507, 57), (501, 4173, 42), (921, 5833, 14))}


@pytest.mark.skipif(not os.path.isdir("/root/reference/datasets"), reason="full ETH/UCY tree only in the build container")
@pytest.mark.parametrize("name", sorted(ALL_COUNTS))
def test_ingest_counts_of_all_fifteen_split_directories(name):
    for part, want in zip(("train", "val", "test"), ALL_COUNTS[name]):
        w = data.load_windows(os.path.join("/root/reference/datasets", name, part), 8, 12, 1, with_non_linear=False)
        assert (len(w), int(w.num_peds.sum()), int(w.num_peds.max())) == want, (name, part)


def test_five_train_splits_from_the_eight_recordings():
    """BASELINE configs[2] ("all five ETH/UCY splits concatenated"): the five leave-one-out train sets built from the
    eight recordings committed under tests/golden/data (seven in eth_train/, biwi_eth_train.txt in train_extra/) have the
    reference's window / pedestrian / largest-crowd counts (measured with its TrajectoryDataset), 11,889
    windows together, and the eth set is the windows of the eth/train directory itself."""
    dirs = [os.path.join(GOLDEN, "data", "eth_train"), os.path.join(GOLDEN, "data", "train_extra")]
    splits = data.load_train_splits(dirs)
    for name, w in splits.items():
        assert (len(w), int(w.num_peds.sum()), int(w.num_peds.max())) == ALL_COUNTS[name][0], name
    assert sum(len(w) for w in splits.values()) == 11889
    eth = data.load_windows(dirs[0], 8, 12, 1, with_non_linear=False)
    assert np.array_equal(eth.seq_rel, splits["eth"].seq_rel) and np.array_equal(eth.num_peds, splits["eth"].num_peds)
    allw = data.concat_windows([splits[s] for s in ("eth", "hotel", "univ", "zara1", "zara2")])
    assert len(allw) == 11889 and allw.seq_start_end[-1][1] == allw.seq_rel.shape[0] == 121709


def test_oracle_reproduces_the_reference_reading_this_repos_checkpoint(tmp_path):
    """N4 fixture (tests/golden/make_golden_ckpt.py: a checkpoint written by trainer.Checkpoint, read back by the REFERENCE
    through its test.py:153-186 flow and evaluated on eth/test): the oracle with the saved weights reproduces the
    reference's V_pred; writing the same state again gives the same val_best.pth contents and reference-style side files."""
    import argparse
    import pickle
    from social_stgcnn_amd.model import social_stgcnn
    from social_stgcnn_amd.trainer import Checkpoint
    g = load_golden("ckpt_roundtrip.npz")
    state = _state(g, "sd/")
    assert list(state.keys()) == list(O.state_dict_keys()) and int(state["st_gcns.0.tcn.0.num_batches_tracked"]) == 1234
    w = _windows("eth_test")
    assert np.array_equal(w.num_peds, g["num_peds"])
    col, worst = 0, 0.0
    with torch.no_grad():
        for i in range(len(w)):
            s, e = w.seq_start_end[i]
            nodes, lap = O.seq_to_graph_np(w.seq_rel[s:e, :, :8].astype(np.float32))
            y = O.social_stgcnn_forward(state, torch.from_numpy(nodes).unsqueeze(0).permute(0, 3, 1, 2),
                                        torch.from_numpy(lap), False)
            worst = max(worst, float(np.abs(y[0].permute(1, 2, 0).numpy() - g["vpred_cat"][:, col:col + (e - s)]).max()))
            col += e - s
    assert col == g["vpred_cat"].shape[1] and worst < 2e-5, worst
    m = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12)
    m.load_state_dict(state)
    d = str(tmp_path / "checkpoint" / "social-stgcnn-roundtrip")
    assert Checkpoint(d, argparse.Namespace(dataset="eth")).record(0, m, 0.5, 0.25)
    again = torch.load(d + "/val_best.pth", weights_only=True)
    for k, v in state.items():
        assert torch.equal(again[k], v) and again[k].dtype == v.dtype, k
    with open(d + "/constant_metrics.pkl", "rb") as fp:
        assert pickle.load(fp) == {"min_val_epoch": int(g["min_val_epoch"]), "min_val_loss": float(g["min_val_loss"])}
