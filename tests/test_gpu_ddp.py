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


def test_dp_pack_and_fold_kernels_equal_their_tensor_statement():
    """stg_dp_pack / stg_dp_fold (three simulated ranks, ragged non-empty scene counts incl. empty scenes) against the
    pure-tensor statement the gloo CPU test runs, and against the plain sequential momentum fold."""
    from social_stgcnn_amd import ops
    from social_stgcnn_amd.trainer import fold_from_pack, pack_rank_slot
    dev = torch.device("cuda", 0)
    world, n_p, n_b, m = 3, 7563, 30, 0.1
    gen = torch.Generator().manual_seed(1)
    before = torch.randn(n_b, generator=gen)
    peds = [torch.tensor([3, 0, 5, 2, 0, 1, 9], dtype=torch.int32), torch.tensor([1, 1, 1], dtype=torch.int32),
            torch.tensor([0, 0, 4, 4, 4, 4, 4, 4, 4, 4, 4], dtype=torch.int32)]
    stats = [torch.randn(int((p > 0).sum()), n_b, generator=gen) for p in peds]
    seq = before.clone()
    packs_dev, packs_ref = [], []
    for r in range(world):
        after = before.clone()
        for srow in stats[r]:
            after = (1 - m) * after + m * srow
            seq = (1 - m) * seq + m * srow
        grad = torch.randn(n_p, generator=gen)
        pk = torch.empty(n_p + world * (n_b + 1), device=dev)
        ops.dp_pack(grad.to(dev), before.to(dev), after.to(dev), peds[r].to(dev), len(peds[r]), m, r, world, pk)
        packs_dev.append(pk)
        packs_ref.append(pack_rank_slot(grad, before, after, int((peds[r] > 0).sum()), m, r, world))
        assert torch.allclose(pk.cpu(), packs_ref[-1], rtol=1e-6, atol=1e-7)
    total_dev = torch.stack(packs_dev).sum(0)
    out = torch.empty(n_b, device=dev)
    ops.dp_fold(total_dev, before.to(dev), m, 0, world, n_p, out)
    ref = fold_from_pack(torch.stack(packs_ref).sum(0), before, m, world, n_p)
    assert torch.allclose(out.cpu(), ref, rtol=1e-6, atol=1e-6)
    assert torch.allclose(out.cpu(), seq, rtol=1e-5, atol=1e-6)
    # num_batches_tracked: rank 1's counters (advanced by its own 3 scenes) take the other ranks' counts from the pack
    own = int((peds[1] > 0).sum())
    nbt = [torch.tensor(own + 100 * k, dtype=torch.int64, device=dev) for k in range(3)]
    ops.dp_fold(total_dev, before.to(dev), m, 1, world, n_p, out, nbt)
    everyone = sum(int((p > 0).sum()) for p in peds)
    assert [int(t) for t in nbt] == [everyone + 100 * k for k in range(3)]
    # the reported loss reduction
    v, w = torch.randn(777, generator=gen), torch.rand(777, generator=gen)
    assert abs(float(ops.weighted_sum(v.to(dev), w.to(dev))) - float((v.double() * w.double()).sum())) < 1e-4
    assert abs(float(ops.weighted_sum(v.to(dev))) - float(v.double().sum())) < 1e-4


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
            # the ranks exchange their scene counts in the step's one all-reduce: the single-process count everywhere
            assert int(got[k]) == int(v) == 16, k
            continue
        upd = float((v - got[k]).abs().max())
        scale = max(1e-6, float(v.abs().max()))
        assert upd < 2e-5 * max(1.0, scale), (k, upd)
