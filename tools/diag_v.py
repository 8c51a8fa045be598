"""Diagnostic (GPU box): gradient errors of the HIP path vs the CPU oracle in fp32 and fp64 for synthetic scenes of V peds."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import stgcnn_oracle as O
from social_stgcnn_amd.model import social_stgcnn
from social_stgcnn_amd.metrics import bivariate_loss
from social_stgcnn_amd import ops
torch.set_num_threads(4)
dev = torch.device("cuda", 0)
v = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rels = []
for i in range(n):
    rng = np.random.default_rng(100 + v + i)
    rel = np.zeros((v, 2, 20), np.float32); rel[:, :, 1:] = np.round(rng.uniform(-0.6, 0.6, (v, 2, 19)), 4).astype(np.float32); rels.append(rel)
torch.manual_seed(v)
m = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12)
state = {k: val.detach().clone() for k, val in m.state_dict().items()}
keys = [k for k, _ in m.named_parameters()]
def oracle(dtype):
    st = {k: (val.to(dtype) if val.is_floating_point() else val.clone()) for k, val in state.items()}
    params = {k: st[k].clone().requires_grad_(True) for k in keys}
    work = dict(st); work.update(params)
    tot = 0; preds = []
    for rel in rels:
        nodes, lap = O.seq_to_graph_np(rel[:, :, :8]); tgt, _ = O.seq_to_graph_np(rel[:, :, 8:])
        l, vp = O.scene_loss(work, torch.from_numpy(nodes).unsqueeze(0).permute(0, 3, 1, 2).to(dtype), torch.from_numpy(lap).to(dtype), torch.from_numpy(tgt).to(dtype), True)
        tot = tot + l; preds.append(vp.detach().double().numpy())
    tot.backward()
    return {k: (None if params[k].grad is None else params[k].grad.double().numpy()) for k in keys}, preds
g32, p32 = oracle(torch.float32); g64, p64 = oracle(torch.float64)
m.to(dev).train()
rel_d = torch.from_numpy(np.stack(rels)).to(dev)
nodes, adj = ops.adj_build(rel_d[..., :8]); tgt = rel_d[..., 8:].permute(0, 3, 1, 2).contiguous()
y, _ = m(nodes.permute(0, 3, 1, 2), adj)
bivariate_loss(y.permute(0, 2, 3, 1), tgt).sum().backward()
print("V_pred max err hip-vs-64 %.2e   torch32-vs-64 %.2e" % (max(np.abs(y[i].detach().permute(1, 2, 0).cpu().double().numpy() - p64[i]).max() for i in range(n)), max(np.abs(p32[i] - p64[i]).max() for i in range(n))))
print("%-32s %10s %12s %12s" % ("param", "|g64|max", "hip-vs-64", "torch32-vs-64"))
for k, p in m.named_parameters():
    if g64[k] is None: continue
    sc = np.abs(g64[k]).max()
    print("%-32s %10.3e %12.3e %12.3e" % (k, sc, np.abs(p.grad.cpu().double().numpy() - g64[k]).max() / max(sc, 1e-30), np.abs(g32[k] - g64[k]).max() / max(sc, 1e-30)))
