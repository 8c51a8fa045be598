import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from social_stgcnn_amd import ops
dev = torch.device("cuda", 0)
for n, v in ((16384, 32), (2048, 32)):
    rel, _ = bench.synth_scenes(256, v, 7)
    rel = torch.from_numpy(np.tile(rel, (n // 256, 1, 1, 1))).to(dev)
    nodes, adj = ops.adj_build(rel)
    ms = bench.time_kernel(torch, lambda: ops.adj_build(rel, out=(nodes, adj)))
    nbytes = n * (64 * v + 32 * v * v + 64 * v)
    print("N=%d V=%d: %.1f us  %.0f GB/s  checksum %.6f" % (n, v, ms * 1e3, nbytes / ms / 1e6, float(adj.double().abs().sum())))
