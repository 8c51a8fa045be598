#!/usr/bin/env python3
"""Golden fixture `train_curve.npz`: the REFERENCE's own train() / vald() driven for several epochs the way
train.py:213-246 drives them -- SGD(lr 0.01), clip_grad_norm_ (train.py:71-73), StepLR (train.py:200,222-223) --
on the 70 eth/test scene-windows.  Per-epoch losses and the final state_dict pin the fused trainer over many
optimizer steps (drift), not just one.

Run in the build container only (the one place /root/reference exists):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_curve.py
Nothing of the reference is copied: the outputs are numbers."""
import argparse
import contextlib
import io
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch

import make_golden as G                     # sets up sys.path for /root/reference, shared helpers
import train as ref_train                   # /root/reference/train.py
import utils as ref_utils                   # /root/reference/utils.py


def main():
    d = os.path.join(G.REF, "datasets", "eth", "test") + "/"
    ds = ref_utils.TrajectoryDataset(d, obs_len=8, pred_len=12, skip=1, norm_lap_matr=True)
    batches = [[t.unsqueeze(0) for t in ds[i]] for i in range(len(ds))]       # DataLoader(batch_size=1, shuffle=False)
    n_sc, bs, lr, clip, epochs, sh_rate = len(batches), 16, 0.01, 0.5, 6, 2
    m = G.new_ref_model(seed=321)
    before = G.sd_to_np(m.state_dict())
    opt = torch.optim.SGD(m.parameters(), lr=lr)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=sh_rate, gamma=0.2)
    targs = argparse.Namespace(batch_size=bs, clip_grad=clip)
    tl, vl = [], []
    for ep in range(epochs):
        with contextlib.redirect_stdout(io.StringIO()):
            tl.append(ref_train.train(ep, m, batches, opt, targs, torch.device("cpu")))
            vl.append(ref_train.vald(ep, m, batches, targs, torch.device("cpu")))
        sched.step()
    out = {"train_loss": np.asarray(tl, np.float64), "val_loss": np.asarray(vl, np.float64),
           "n_scenes": np.int64(n_sc), "batch_size": np.int64(bs), "lr": np.float64(lr),
           "clip_grad": np.float64(clip), "epochs": np.int64(epochs), "lr_sh_rate": np.int64(sh_rate)}
    for k, v in before.items():
        out["before/" + k] = v
    for k, v in G.sd_to_np(m.state_dict()).items():
        out["after/" + k] = v
    np.savez(os.path.join(HERE, "train_curve.npz"), **out)
    print("train", tl)
    print("val  ", vl)


if __name__ == "__main__":
    main()
