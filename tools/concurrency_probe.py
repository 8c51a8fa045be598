"""Does running two half-batch steps concurrently (two streams) beat one full-batch step?  Probe for a
two-chain step design: two independent models/trainers, each captured on N/2 scenes, replayed on two streams."""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
from social_stgcnn_amd import ops
from social_stgcnn_amd.model import social_stgcnn
from social_stgcnn_amd.trainer import Trainer

dev = torch.device("cuda", 0)
V = 32


def make(n, seed):
    obs_rel, target = bench.synth_scenes(n, V, seed)
    nodes, adj = ops.adj_build(torch.from_numpy(obs_rel).to(dev))
    x, tgt = nodes.permute(0, 3, 1, 2), torch.from_numpy(target).to(dev)
    w = torch.full((n,), 1.0 / n, device=dev)
    torch.manual_seed(0)
    m = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12).to(dev).train()
    tr = Trainer(m, lr=0.01)
    return tr.capture(x, adj, tgt, None, w), (x, adj, tgt, w, m, tr)


def timeit(fn, steps=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


full, keep0 = make(2048, 1)
print("one graph, 2048 scenes: %.3f ms" % timeit(full))
for parts in (2, 4):
    n = 2048 // parts
    reps, keeps, streams = [], [], []
    for i in range(parts):
        r, k = make(n, 10 + i)
        reps.append(r); keeps.append(k); streams.append(torch.cuda.Stream())

    def both():
        for r, s in zip(reps, streams):
            with torch.cuda.stream(s):
                r()
    print("%d graphs x %d scenes on %d streams: %.3f ms" % (parts, n, parts, timeit(both)))

    def serial():
        for r in reps:
            r()
    print("%d graphs x %d scenes, one stream: %.3f ms" % (parts, n, timeit(serial)))
