"""Diagnostic: one real eth/train group (device gather -> adj_build -> fused forward + loss + backward) against the CPU
oracle run scene by scene: per-scene V_pred / loss errors and per-parameter gradient errors.
   python tools/group_vs_oracle.py [first window of the shuffled order] [scenes]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import stgcnn_oracle as O                       # noqa: E402
from social_stgcnn_amd import data, ops                     # noqa: E402
from social_stgcnn_amd.dataset import DeviceWindows         # noqa: E402
from social_stgcnn_amd.metrics import bivariate_loss        # noqa: E402
from social_stgcnn_amd.model import social_stgcnn           # noqa: E402

dev = torch.device("cuda", 0)
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 512
win = data.load_windows(os.path.join(ROOT, "tests", "golden", "data", "eth_train"), 8, 12, 1, with_non_linear=False)
ds = DeviceWindows(win, dev)
gen = torch.Generator(device=dev).manual_seed(0)
order = ds.shuffled_order(gen)
idx = order[lo:lo + cnt].contiguous()
obs_rel, target, peds = ds.gather(idx, v_pad=(ds.v_max + 3) & ~3)
nodes, adj = ops.adj_build(obs_rel, peds)
torch.manual_seed(0)
m = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12).to(dev).train()
state = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
y, _ = m(nodes.permute(0, 3, 1, 2), adj, peds)
losses = bivariate_loss(y.permute(0, 2, 3, 1), target, peds)
losses.sum().backward()
# oracle
keys = [k for k, _ in m.named_parameters()]
params = {k: state[k].clone().requires_grad_(True) for k in keys}
work = dict(state)
work.update(params)
pc = peds.cpu().numpy()
xc, ac, tc = nodes.permute(0, 3, 1, 2).cpu(), adj.cpu(), target.cpu()
yc, lc = y.detach().cpu(), losses.detach().cpu()
tot = 0
worst_y, worst_l, worst_scene = 0.0, 0.0, -1
for i in range(cnt):
    v = int(pc[i])
    l, vp = O.scene_loss(work, xc[i:i + 1, :, :, :v], ac[i, :, :v, :v], tc[i, :, :v], True)
    tot = tot + l
    ey = float((yc[i, :, :, :v].permute(1, 2, 0) - vp.detach()).abs().max())
    el = abs(float(lc[i]) - float(l))
    if ey > worst_y:
        worst_y, worst_scene = ey, i
    worst_l = max(worst_l, el)
tot.backward()
print("scenes %d (V up to %d): worst |V_pred| error %.2e (scene %d, %d peds), worst loss error %.2e"
      % (cnt, int(pc.max()), worst_y, worst_scene, int(pc[worst_scene]), worst_l))
for name, p in m.named_parameters():
    ref = params[name].grad
    if ref is None:
        continue
    scale = max(1e-3, float(ref.abs().max()))
    err = float((p.grad.cpu() - ref).abs().max()) / scale
    flag = "  <<<<" if err > 1e-3 else ""
    print("  %-32s rel err %.2e%s" % (name, err, flag))
