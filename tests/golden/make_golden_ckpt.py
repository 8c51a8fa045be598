#!/usr/bin/env python3
"""Golden fixture for N4 (checkpoint files): a checkpoint WRITTEN BY THIS REPO (social_stgcnn_amd.trainer.Checkpoint:
val_best.pth / args.pkl / metrics.pkl / constant_metrics.pkl, train.py:202-246) is read back by the REFERENCE the way
its test.py:153-186 does -- pickle.load(args.pkl), pickle.load(constant_metrics.pkl), model.social_stgcnn(**args),
model.load_state_dict(torch.load(val_best.pth)) -- and evaluated with its test() on eth/test (torch.manual_seed(0)).
Build container only (/root/reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ckpt.py

Writes ckpt_roundtrip.npz (DATA only): the state_dict that was saved (sd/<key>), V_pred of the reference model for all 70
eth/test windows (concatenated along the pedestrians), its ADE / FDE and per-pedestrian best-of-20 errors, the args and
constant metrics the reference read from the files."""
import argparse
import contextlib
import io
import os
import pickle
import sys
import tempfile

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)

import numpy as np
import torch

import model as ref_model          # /root/reference/model.py
import utils as ref_utils          # /root/reference/utils.py
import metrics as ref_metrics      # /root/reference/metrics.py

torch.set_num_threads(1)


def trained_like_state():
    """a state that is none of the shipped checkpoints and has every kind of entry off its default: seeded init, the
    parameters and BatchNorm buffers perturbed deterministically, num_batches_tracked counted up"""
    from social_stgcnn_amd.model import social_stgcnn
    torch.manual_seed(11)
    m = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12)
    g = torch.Generator().manual_seed(12)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=g))
        for name, b in m.named_buffers():
            if name.endswith("running_mean"):
                b.add_(0.1 * torch.randn(b.shape, generator=g))
            elif name.endswith("running_var"):
                b.mul_(1.0 + 0.2 * torch.rand(b.shape, generator=g))
            elif name.endswith("num_batches_tracked"):
                b.fill_(1234)
    return m


def main():
    from social_stgcnn_amd.trainer import Checkpoint
    m = trained_like_state()
    # the reference's own argument set (train.py:127-156), as its args.pkl holds it
    args = argparse.Namespace(input_size=2, output_size=5, n_stgcnn=1, n_txpcnn=5, kernel_size=3, obs_seq_len=8,
                              pred_seq_len=12, dataset="eth", batch_size=128, num_epochs=250, clip_grad=None, lr=0.01,
                              lr_sh_rate=150, use_lrschd=False, tag="social-stgcnn-roundtrip")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        exp_path = os.path.join(tmp, "checkpoint", args.tag)
        ck = Checkpoint(exp_path, args)
        assert ck.record(0, m, 0.5, 0.25)
        assert sorted(os.listdir(exp_path)) == ["args.pkl", "constant_metrics.pkl", "metrics.pkl", "val_best.pth"]
        # ---- the reference's reader (test.py:153-186), on files this repo wrote
        model_path = exp_path + "/val_best.pth"
        with open(exp_path + "/args.pkl", "rb") as f:
            rargs = pickle.load(f)
        with open(exp_path + "/constant_metrics.pkl", "rb") as f:
            cm = pickle.load(f)
        device = torch.device("cpu")
        model = ref_model.social_stgcnn(n_stgcnn=rargs.n_stgcnn, n_txpcnn=rargs.n_txpcnn, output_feat=rargs.output_size,
                                        seq_len=rargs.obs_seq_len, kernel_size=rargs.kernel_size,
                                        pred_seq_len=rargs.pred_seq_len).to(device)
        model.load_state_dict(torch.load(model_path, map_location=device))          # strict: all 40 keys
        saved = torch.load(model_path, map_location="cpu", weights_only=True)
    for k, v in saved.items():
        out["sd/" + k] = v.numpy().copy()
        assert torch.equal(v, m.state_dict()[k]), k
    # ---- the reference's test() (test.py:18-127) on its eth/test data
    src = open(os.path.join(REF, "test.py")).read()
    head = src.split("paths = ['./checkpoint/*social-stgcnn*']")[0]
    ns = {}
    exec(compile(head, os.path.join(REF, "test.py"), "exec"), ns)
    d = os.path.join(REF, "datasets", rargs.dataset, "test")
    ds = ref_utils.TrajectoryDataset(d + "/", obs_len=rargs.obs_seq_len, pred_len=rargs.pred_seq_len, skip=1,
                                     norm_lap_matr=True)
    batches = [[t.unsqueeze(0) for t in ds[i]] for i in range(len(ds))]
    model.eval()
    ns["model"] = model
    ns["loader_test"] = batches
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        ade_, fde_, raw = ns["test"](KSTEPS=20)
    per_ade, per_fde = [], []
    for step in range(1, len(batches) + 1):
        tgt = raw[step]["trgt"]
        for n in range(tgt.shape[1]):
            per_ade.append(min(ref_metrics.ade([p[:, n:n + 1, :]], [tgt[:, n:n + 1, :]], [1]) for p in raw[step]["pred"]))
            per_fde.append(min(ref_metrics.fde([p[:, n:n + 1, :]], [tgt[:, n:n + 1, :]], [1]) for p in raw[step]["pred"]))
    vps = []
    with torch.no_grad():
        for b in batches:
            y, _ = model(b[6].permute(0, 3, 1, 2), b[7].squeeze())
            vps.append(y.permute(0, 2, 3, 1).squeeze(0).numpy().reshape(12, -1, 5))
    out["vpred_cat"] = np.concatenate(vps, axis=1)
    out["num_peds"] = np.asarray([e - s for s, e in ds.seq_start_end], dtype=np.int32)
    out["ade"], out["fde"] = np.float64(ade_), np.float64(fde_)
    out["per_ped_ade"], out["per_ped_fde"] = np.asarray(per_ade), np.asarray(per_fde)
    out["min_val_epoch"], out["min_val_loss"] = np.int64(cm["min_val_epoch"]), np.float64(cm["min_val_loss"])
    out["args_dataset"] = np.asarray(rargs.dataset)
    np.savez(os.path.join(HERE, "ckpt_roundtrip.npz"), **out)
    print("reference read this repo's checkpoint: %d windows, ADE %.6f FDE %.6f, constant metrics %s"
          % (len(ds), ade_, fde_, cm))


if __name__ == "__main__":
    main()
