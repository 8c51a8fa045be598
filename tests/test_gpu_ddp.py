"""GPU (-m gpu): two ranks sharing ONE GPU over gloo == one rank on the concatenated batch
(gradient all-reduce + exact BatchNorm fold + SGD), eager and hipGraph-captured."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _data(dev, n, v, seed):
    import bench
    from social_stgcnn_amd import ops
    obs_rel, target = bench.synth_scenes(n, v, seed)
    nodes, adj = ops.adj_build(torch.from_numpy(obs_rel).to(dev))
    return nodes.permute(0, 3, 1, 2).contiguous(), adj, torch.from_numpy(target).to(dev)


def _worker(rank, world, port, out_path, captured):
    import torch.distributed as dist
    from social_stgcnn_amd.model import social_stgcnn
    from social_stgcnn_amd.trainer import Trainer, broadcast_module
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    x, adj, tgt = _data(dev, 16, 12, 5)
    lo, hi = rank * 8, rank * 8 + 8
    x, adj, tgt = x[lo:hi].contiguous(), adj[lo:hi].contiguous(), tgt[lo:hi].contiguous()
    torch.manual_seed(3 + rank)                      # different init per rank: the broadcast must fix it
    m = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12).to(dev).train()
    broadcast_module(m)
    tr = Trainer(m, lr=0.05)
    w = torch.full((8,), 1.0 / 16, device=dev)
    if captured:
        step = tr.capture(x, adj, tgt, None, w, warmup=0)
        step()
    else:
        tr.step(x, adj, tgt, None, w)
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({k: v.cpu() for k, v in m.state_dict().items()}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("captured", (False, True))
def test_two_ranks_equal_one_rank_on_concatenated_batch(tmp_path, captured):
    import torch.multiprocessing as mp
    from social_stgcnn_amd.model import social_stgcnn
    from social_stgcnn_amd.trainer import Trainer
    dev = torch.device("cuda", 0)
    x, adj, tgt = _data(dev, 16, 12, 5)
    torch.manual_seed(3)                             # rank 0's initialisation
    m = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12).to(dev).train()
    Trainer(m, lr=0.05).step(x, adj, tgt, None, torch.full((16,), 1.0 / 16, device=dev))
    ref = {k: v.cpu() for k, v in m.state_dict().items()}
    out_path = str(tmp_path / "rank0.pt")
    ctx = mp.get_context("spawn")
    port = 29600 + os.getpid() % 1000 + (1 if captured else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out_path, captured)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = torch.load(out_path, weights_only=True)
    for k, v in ref.items():
        if "num_batches" in k:
            # each rank counts its own 8 forwards (the reference counts per process too)
            assert int(got[k]) in (8, 16), k
            continue
        upd = float((v - got[k]).abs().max())
        scale = max(1e-6, float(v.abs().max()))
        assert upd < 2e-5 * max(1.0, scale), (k, upd)
