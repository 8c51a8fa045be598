"""Diagnostic: run-to-run determinism of the fused step on real ragged batches (a data race shows up as a difference
between two runs from identical state), and a long training loop watching for the first non-finite value.
   python tools/race_check.py [batch] [repeats] [long_steps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                  # noqa: E402
from social_stgcnn_amd import ops                             # noqa: E402
from social_stgcnn_amd.metrics import bivariate_loss          # noqa: E402
from social_stgcnn_amd.model import social_stgcnn             # noqa: E402
from social_stgcnn_amd.trainer import Trainer                 # noqa: E402

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
long_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
raw, _ = bench.eth_train_batches(n, 2, seed=1)
sets = []
for obs_rel, target, counts in raw:
    rel = torch.from_numpy(obs_rel).to(dev)
    peds = torch.from_numpy(counts.astype(np.int32)).to(dev)
    nodes, adj = ops.adj_build(rel, peds)
    sets.append((nodes.permute(0, 3, 1, 2), adj, torch.from_numpy(target).to(dev), peds))
torch.manual_seed(0)
model = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12).to(dev).train()
state = {k: v.clone() for k, v in model.state_dict().items()}


def run(x, adj, tgt, peds):
    model.load_state_dict(state)
    model.zero_grad(set_to_none=True)
    y, _ = model(x, adj, peds)
    l = bivariate_loss(y.permute(0, 2, 3, 1), tgt, peds)
    l.sum().backward()
    g = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])
    return y.detach().clone(), l.detach().clone(), g.clone()


bad = 0
for si, (x, adj, tgt, peds) in enumerate(sets):
    y0, l0, g0 = run(x, adj, tgt, peds)
    print("set %d: V=%d finite y %s l %s g %s" % (si, x.shape[3], bool(torch.isfinite(y0).all()), bool(torch.isfinite(l0).all()),
                                                   bool(torch.isfinite(g0).all())))
    for r in range(reps):
        y1, l1, g1 = run(x, adj, tgt, peds)
        dy = int((y1 != y0).sum())
        dl = int((l1 != l0).sum())
        dg = float((g1 - g0).abs().max() / g0.abs().max())
        if dy or dl or dg > 1e-5:
            bad += 1
            sc = torch.nonzero((y1 != y0).flatten(1).any(1)).flatten().tolist()[:8]
            print("  rep %d: y differs in %d values (scenes %s, peds %s), losses differ %d, grad rel diff %.2e"
                  % (r, dy, sc, [int(peds[s]) for s in sc], dl, dg))
print("determinism: %d bad repetitions" % bad)

if long_steps:
    model.load_state_dict(state)
    trainer = Trainer(model, lr=0.01)
    w = torch.full((n,), 1.0 / n, device=dev)
    steps = [trainer.capture(x, adj, tgt, peds, w) for (x, adj, tgt, peds) in sets]
    for i in range(long_steps):
        out = steps[i % 2]()
        if i % 100 == 99 or i == long_steps - 1:
            flat = model.flat_parameters()
            ok = bool(torch.isfinite(flat).all())
            print("step %d loss %.6f params finite %s |p|max %.3f" % (i + 1, float(out[0]), ok, float(flat.abs().max())))
            if not ok:
                break
