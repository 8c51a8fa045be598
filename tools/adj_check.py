"""Diagnostic: stg_adj_build on a ragged batch padded to V against the oracle, for several paddings."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import stgcnn_oracle as O                       # noqa: E402
from social_stgcnn_amd import ops                           # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
for V in (57, 60, 64, 12, 36):
    counts = [V, 2, 5, min(V, 33), 1, min(V, 17), V - 1, 3]
    rel = np.zeros((len(counts), V, 2, 8), np.float32)
    for i, c in enumerate(counts):
        rel[i, :c, :, 1:] = np.round(rng.uniform(-0.6, 0.6, (c, 2, 7)), 4)
    nodes, adj = ops.adj_build(torch.from_numpy(rel).to(dev), torch.tensor(counts, dtype=torch.int32, device=dev))
    worst = 0.0
    for i, c in enumerate(counts):
        _, lap = O.seq_to_graph_np(rel[i, :c])
        got = adj[i].cpu().numpy()
        worst = max(worst, float(np.abs(got[:, :c, :c] - lap).max()))
        assert np.all(got[:, c:, :] == 0) and np.all(got[:, :, c:] == 0), (V, i)
    print("V=%d worst |adj - oracle| %.2e" % (V, worst))
