"""Diagnostic (STG_STAMPS=1): per-phase wave timelines of txp_fwd_wave_kernel."""
import os, sys
os.environ["STG_STAMPS"] = "1"
os.environ["STG_USE_DIAG_LIB"] = "1"        # stamps exist only in the diagnostic build (make -C csrc DIAG=1)
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from social_stgcnn_amd import ops
from social_stgcnn_amd.model import social_stgcnn
dev = torch.device("cuda", 0)
n, v = int(os.environ.get("STAMPS_N", "2048")), int(os.environ.get("STAMPS_V", "32"))
ops.OPTIONS["f32_mfma"] = True          # the stamps live in the fp32-MFMA wave kernel (txp_fwd_wave_kernel)
ops.OPTIONS["wave_path"] = True
obs_rel, target = bench.synth_scenes(n, v, 1)
nodes, adj = ops.adj_build(torch.from_numpy(obs_rel).to(dev))
x = nodes.permute(0, 3, 1, 2)
torch.manual_seed(0)
m = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12).to(dev).train()
if os.environ.get("STAMPS_EVAL"):
    m.eval()
    with torch.no_grad():
        for _ in range(3): m(x, adj)
else:
    for _ in range(3): m(x, adj)
torch.cuda.synchronize()
scr = m._last_fwd_scratch
off = ((n * 3 * 8 * v + 3) & ~3) + 4 + ((n + v + 2 + 3) & ~3)     # aggregated input, pad, scene order
st = scr[off: off + n * 32].cpu().numpy().view(np.uint64).reshape(n, 16).astype(np.int64)
t0 = st[:, 0].min()
print("kernel span (cycles): %d   scenes: %d" % (st[:, 8].max() - t0, n))
d = np.diff(st[:, :9], axis=1)
names = ["dma+w0 wait", "layer0", "layer1", "layer2", "layer3", "(unused)", "(unused)", "out layer"]
for k in range(8):
    if k in (5, 6): continue
    col = d[:, k] if k < 5 else None
names2 = ["st_gcn block", "a0 save+l0", "layer1", "layer2", "layer3", "out"]   # (fine stamps 9..14 are shared: block phases / layer 1)
seg = np.stack([st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2], st[:, 4] - st[:, 3], st[:, 5] - st[:, 4], st[:, 8] - st[:, 5]], 1)
for k, nm in enumerate(names2):
    print("%-12s median %7d  p10 %7d  p90 %7d cycles" % (nm, np.median(seg[:, k]), np.percentile(seg[:, k], 10), np.percentile(seg[:, k], 90)))
life = st[:, 8] - st[:, 0]
print("scene total  median %d cycles; start spread: p50 %d p90 %d max %d" % (np.median(life), np.percentile(st[:, 0] - t0, 50), np.percentile(st[:, 0] - t0, 90), (st[:, 0] - t0).max()))

# st_gcn block phases (stamps 9..14 inside stgcn_block_fwd, wave mode)
if st[:, 14].max() > 0:
    cols = [0, 9, 10, 11, 12, 13, 14, 1]
    nm = ["ptab+loads+g", "BN1 stats", "bn+prelu", "tconv+save", "BN2/BNr stats", "header", "output->plane"]
    for k in range(7):
        d_ = st[:, cols[k + 1]] - st[:, cols[k]]
        print("  %-16s median %7d  p10 %7d  p90 %7d cycles" % (nm[k], np.median(d_), np.percentile(d_, 10), np.percentile(d_, 90)))

# fine stamps of layer 1 (diagnostic build): 2 -> 12 (pointer setup + saved-border zeroing), 12 -> 9 (tile loop),
# 9 -> 10 (weight fetch issue), 10 -> 11 (border row zeroing)
if st[:, 12].max() > 0:
    fine = np.stack([st[:, 12] - st[:, 2], st[:, 9] - st[:, 12], st[:, 10] - st[:, 9], st[:, 11] - st[:, 10]], 1)
    for k, nm in enumerate(["l1 setup", "l1 tiles", "l1 wfetch", "l1 zero row"]):
        print("%-12s median %7d  p10 %7d  p90 %7d cycles" % (nm, np.median(fine[:, k]), np.percentile(fine[:, k], 10), np.percentile(fine[:, k], 90)))
