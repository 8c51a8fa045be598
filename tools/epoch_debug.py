"""Diagnostic: device-resident eth/train epochs (EpochRunner) with a health check after every group: the first group
whose losses / outputs / parameters are not finite is reported with its scenes.
   python tools/epoch_debug.py [batch] [epochs]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from social_stgcnn_amd import data                          # noqa: E402
from social_stgcnn_amd.dataset import DeviceWindows, EpochRunner   # noqa: E402
from social_stgcnn_amd.model import social_stgcnn           # noqa: E402
from social_stgcnn_amd.trainer import Trainer, group_bounds  # noqa: E402

dev = torch.device("cuda", 0)
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_ep = int(sys.argv[2]) if len(sys.argv) > 2 else 8
v_pad = int(sys.argv[3]) if len(sys.argv) > 3 and int(sys.argv[3]) > 0 else None
eager = len(sys.argv) > 4 and sys.argv[4] == "eager"
win = data.load_windows(os.path.join(ROOT, "tests", "golden", "data", "eth_train"), 8, 12, 1, with_non_linear=False)
torch.manual_seed(0)
model = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12).to(dev).train()
trainer = Trainer(model, lr=0.01)
ds = DeviceWindows(win, dev)
runner = EpochRunner(trainer, ds, bs, v_pad=v_pad)
print("batch", bs, "padded V", runner.obs_rel.shape[1])
gen = torch.Generator(device=dev).manual_seed(0)
done = False
for ep in range(n_ep):
    order = ds.shuffled_order(gen)
    lo, tot = 0, 0.0
    for gi, b in enumerate(group_bounds(order.numel(), bs)):
        cnt = b + 1 - lo
        if cnt not in runner._replays:
            runner.index[:cnt].copy_(order[lo:b + 1])
        before = model.flat_parameters().clone()
        if eager:
            from social_stgcnn_amd import ops
            from social_stgcnn_amd.trainer import group_weights
            idx = order[lo:b + 1].contiguous()
            obs_rel, target, peds_g = ds.gather(idx, v_pad=runner.obs_rel.shape[1])
            nodes, adj = ops.adj_build(obs_rel, peds_g)
            total, losses, y = trainer.step(nodes.permute(0, 3, 1, 2), adj, target, peds_g, group_weights(cnt, bs, dev))
            runner.peds[:cnt].copy_(peds_g)
            index = idx
        else:
            replay, index, _ = runner._group(cnt)
            index.copy_(order[lo:b + 1])
            total, losses, y = replay()
        torch.cuda.synchronize()
        gnorm = float((model.flat_parameters() - before).norm())
        print("  ep %d group %d n=%d loss*bs %.4f  |dp| %.4f  max scene loss %.3f" % (ep, gi, cnt, float(total) * bs, gnorm, float(losses.max())))
        flat = model.flat_parameters()
        okl, oky, okp = bool(torch.isfinite(losses).all()), bool(torch.isfinite(y).all()), bool(torch.isfinite(flat).all())
        tot += float(total)
        if not (okl and oky and okp):
            peds = runner.peds[:cnt]
            badl = torch.nonzero(~torch.isfinite(losses)).flatten().tolist()
            bady = torch.nonzero(~torch.isfinite(y).flatten(1).all(1)).flatten().tolist()
            print("epoch %d group %d (%d scenes): losses finite %s, y finite %s, params finite %s (before: %s)"
                  % (ep, gi, cnt, okl, oky, okp, bool(torch.isfinite(before).all())))
            print("  bad loss scenes", badl[:16], "peds", [int(peds[i]) for i in badl[:16]])
            print("  bad y scenes", bady[:16], "peds", [int(peds[i]) for i in bady[:16]])
            for i in (bady or badl)[:2]:
                yy = y[i]
                print("  scene %d window %d peds %d: y nonfinite count %d, per-feature %s" % (
                    i, int(index[i]), int(peds[i]), int((~torch.isfinite(yy)).sum()),
                    [int((~torch.isfinite(yy[f])).sum()) for f in range(5)]))
                print("   |y| max per feature (finite part)", [float(torch.nan_to_num(yy[f], 0, 0, 0).abs().max()) for f in range(5)])
            done = True
            break
        lo = b + 1
    print("epoch %d loss %.6f |p|max %.3f" % (ep, tot / order.numel(), float(model.flat_parameters().abs().max())))
    if done:
        break
