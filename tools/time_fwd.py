"""Diagnostic: forward-only timings (inference = no activation saves vs training = saves)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from social_stgcnn_amd import ops
from social_stgcnn_amd.model import social_stgcnn
dev = torch.device("cuda", 0)
n, v = 2048, int(os.environ.get("PEDS", "32"))
obs_rel, target = bench.synth_scenes(n, v, 1)
nodes, adj = ops.adj_build(torch.from_numpy(obs_rel).to(dev))
x = nodes.permute(0, 3, 1, 2)
torch.manual_seed(0)
m = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12).to(dev)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - a) / it * 1e3
m.eval()
with torch.no_grad():
    print("eval no_grad fwd ms: %.3f" % t(lambda: m(x, adj)))
m.train()
print("train fwd (saves) ms: %.3f" % t(lambda: m(x, adj)))
