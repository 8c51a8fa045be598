"""Diagnostic (GPU box): per-parameter gradient errors of the HIP path vs the CPU oracle in fp32 and fp64
for the first reference training group (16 eth/test scenes)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import stgcnn_oracle as O
from social_stgcnn_amd.model import social_stgcnn
from social_stgcnn_amd.metrics import bivariate_loss
from social_stgcnn_amd import ops
torch.set_num_threads(1)
dev = torch.device("cuda", 0)
g = np.load(os.path.join(ROOT, "tests/golden/train_loop.npz")); e = np.load(os.path.join(ROOT, "tests/golden/eth_test_windows.npz"))
state = {k[7:]: torch.from_numpy(np.array(g[k])) for k in g.files if k.startswith("before/")}
keys = [k for k in state if "running" not in k and "num_batches" not in k]
starts = np.concatenate([[0], np.cumsum(e["num_peds"])])
nsc = int(sys.argv[1]) if len(sys.argv) > 1 else 16
def oracle(dtype):
    st = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in state.items()}
    params = {k: st[k].clone().requires_grad_(True) for k in keys}
    work = dict(st); work.update(params)
    tot = 0
    for i in range(nsc):
        rel = e["seq_rel"][starts[i]:starts[i + 1]]
        nodes, lap = O.seq_to_graph_np(rel[:, :, :8]); tgt, _ = O.seq_to_graph_np(rel[:, :, 8:])
        l, _ = O.scene_loss(work, torch.from_numpy(nodes).unsqueeze(0).permute(0, 3, 1, 2).to(dtype), torch.from_numpy(lap).to(dtype), torch.from_numpy(tgt).to(dtype), True)
        tot = tot + l
    tot.backward()
    return {k: (None if params[k].grad is None else params[k].grad.double().numpy()) for k in keys}
g32, g64 = oracle(torch.float32), oracle(torch.float64)
m = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12)
m.load_state_dict(state); m.to(dev).train()
peds = e["num_peds"][:nsc]; vmax = int(peds.max())
rel = np.zeros((nsc, vmax, 2, 20), np.float32)
for i in range(nsc): rel[i, :peds[i]] = e["seq_rel"][starts[i]:starts[i + 1]]
rel_d = torch.from_numpy(rel).to(dev); peds_d = torch.from_numpy(peds.astype(np.int32)).to(dev)
nodes, adj = ops.adj_build(rel_d[..., :8], peds_d)
tgt = rel_d[..., 8:].permute(0, 3, 1, 2).contiguous()
y, _ = m(nodes.permute(0, 3, 1, 2), adj, peds_d)
bivariate_loss(y.permute(0, 2, 3, 1), tgt, peds_d).sum().backward()
print("%-32s %10s %12s %12s" % ("param", "|g64|max", "hip-vs-64", "torch32-vs-64"))
for k, p in m.named_parameters():
    if g64[k] is None: continue
    sc = np.abs(g64[k]).max()
    print("%-32s %10.3e %12.3e %12.3e" % (k, sc, np.abs(p.grad.cpu().double().numpy() - g64[k]).max() / max(sc, 1e-30), np.abs(g32[k] - g64[k]).max() / max(sc, 1e-30)))
