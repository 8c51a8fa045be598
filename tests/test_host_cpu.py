"""CPU: host logic -- ingest against the reference TrajectoryDataset fixture, group semantics,
and the data-parallel pieces on a world-size-2 gloo group."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden


def test_ingest_matches_reference_dataset():
    """N1: load_windows on the raw eth/test file == the reference TrajectoryDataset (fixture)."""
    from social_stgcnn_amd import data
    g = load_golden("eth_test_windows.npz")
    w = data.load_windows(os.path.join(GOLDEN, "data", "eth_test"), 8, 12, 1)
    assert len(w) == 70 and int(w.num_peds.sum()) == 181 and int(w.num_peds.max()) == 5
    assert np.array_equal(w.num_peds, g["num_peds"])
    assert np.array_equal(w.seq.astype(np.float32), g["seq"])
    assert np.array_equal(w.seq_rel.astype(np.float32), g["seq_rel"])
    assert np.array_equal(w.non_linear.astype(np.float32), g["non_linear"])
    assert np.array_equal(w.loss_mask.astype(np.float32), g["loss_mask"])
    assert w.max_peds_in_frame == int(g["max_peds_in_frame"])
    obs_rel, pred_rel, obs_abs, pred_abs, peds = data.pad_batch(w, np.arange(4))
    assert obs_rel.shape == (4, 8, int(peds.max()), 2) and pred_rel.shape[1] == 12
    assert np.array_equal(obs_rel[0, :, :peds[0]], g["v_obs0"])          # V_obs of the first window
    assert np.array_equal(pred_rel[0, :, :peds[0]], g["v_tr0"])
    assert np.all(obs_rel[0, :, peds[0]:] == 0)


def test_group_semantics():
    from social_stgcnn_amd.trainer import group_bounds, group_weights
    from oracle import stgcnn_oracle as O
    for n, b in ((40, 16), (2785, 128), (256, 128), (5, 8), (130, 128)):
        assert group_bounds(n, b) == O.group_boundaries(n, b)
    w = group_weights(16, 16)
    assert float(w[-1]) == 0 and np.allclose(w[:-1].numpy(), 1 / 16)


def _dp_worker(rank, world, port, out):
    import torch.distributed as dist
    from social_stgcnn_amd.trainer import (allreduce_flat, broadcast_module, fold_bn_across_ranks, fold_from_pack,
                                           pack_rank_slot)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    # gradient all-reduce: one flat buffer, sum over ranks
    g = torch.arange(7563, dtype=torch.float32) * (rank + 1)
    allreduce_flat(g)
    assert torch.equal(g, torch.arange(7563, dtype=torch.float32) * sum(r + 1 for r in range(world)))
    # BatchNorm fold: every rank folds its own scenes, the cross-rank fold equals one sequential pass
    m = 0.1
    before = torch.linspace(-1, 1, 30)
    gen = torch.Generator().manual_seed(7)
    stats = torch.randn(world, 5 + 3, 30, generator=gen)              # same on all ranks
    n_local = 5 + 3 * rank                                            # ragged shard sizes
    after = before.clone()
    for i in range(n_local):
        after = (1 - m) * after + m * stats[rank, i]
    folded = fold_bn_across_ranks(before, after, n_local, m)
    ref = before.clone()
    for r in range(world):
        for i in range(5 + 3 * r):
            ref = (1 - m) * ref + m * stats[r, i]
    assert torch.allclose(folded, ref, rtol=1e-5, atol=1e-6), float((folded - ref).abs().max())
    # the trainer's form: gradient, BatchNorm contribution and scene count in ONE all-reduced buffer (what stg_dp_pack /
    # stg_dp_fold do on the device), no count exchange beforehand
    grad = torch.arange(40, dtype=torch.float32) * (rank + 1)
    pack = pack_rank_slot(grad, before, after, n_local, m, rank, world)
    assert pack.numel() == 40 + world * 31
    dist.all_reduce(pack)
    assert torch.equal(pack[:40], torch.arange(40, dtype=torch.float32) * sum(r + 1 for r in range(world)))
    folded2 = fold_from_pack(pack, before, m, world, 40)
    assert torch.allclose(folded2, ref, rtol=1e-5, atol=1e-6), float((folded2 - ref).abs().max())
    # initial broadcast
    lin = torch.nn.Linear(4, 3)
    broadcast_module(lin)
    w = [torch.empty_like(lin.weight) for _ in range(world)]
    dist.all_gather(w, lin.weight.data)
    assert all(torch.equal(w[0], t) for t in w)
    dist.barrier()
    dist.destroy_process_group()
    out.put(rank)


def test_data_parallel_pieces_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(out.get(timeout=5) for _ in range(2)) == [0, 1]


def test_checkpoint_files_follow_the_reference_layout(tmp_path):
    """N4 (train.py:202-246): args.pkl / metrics.pkl / constant_metrics.pkl / val_best.pth; the saved state_dict has
    the reference's 40 keys, loads with weights_only=True and round-trips into a fresh model."""
    import argparse
    import pickle
    from oracle import stgcnn_oracle as o
    from social_stgcnn_amd.model import social_stgcnn
    from social_stgcnn_amd.trainer import Checkpoint, load_checkpoint
    torch.manual_seed(0)
    m = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12)
    d = str(tmp_path / "checkpoint" / "tag")
    ck = Checkpoint(d, argparse.Namespace(n_stgcnn=1, n_txpcnn=5, lr=0.01, tag="tag"))
    assert ck.record(0, m, 1.5, 2.0) is True
    with torch.no_grad():
        m.tpcnn_ouput.bias.add_(1.0)
    assert ck.record(1, m, 1.4, 2.5) is False                      # worse validation loss: no save
    assert ck.record(2, m, 1.3, 1.0) is True
    with open(d + "/metrics.pkl", "rb") as fp:
        assert pickle.load(fp) == {"train_loss": [1.5, 1.4, 1.3], "val_loss": [2.0, 2.5, 1.0]}
    with open(d + "/constant_metrics.pkl", "rb") as fp:
        assert pickle.load(fp) == {"min_val_epoch": 2, "min_val_loss": 1.0}
    with open(d + "/args.pkl", "rb") as fp:
        assert pickle.load(fp).n_txpcnn == 5
    sd = torch.load(d + "/val_best.pth", weights_only=True)
    assert list(sd.keys()) == list(o.state_dict_keys())
    m2 = load_checkpoint(social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3,
                                       pred_seq_len=12), d + "/val_best.pth")
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k


def test_step_lr_schedule_matches_torch():
    from social_stgcnn_amd.model import social_stgcnn
    from social_stgcnn_amd.trainer import Trainer
    m = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12)
    tr = Trainer(m, lr=0.01, lr_sh_rate=3)
    opt = torch.optim.SGD(m.parameters(), lr=0.01)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=3, gamma=0.2)
    for _ in range(10):
        opt.step()
        sched.step()
        assert abs(tr.scheduler_step() - opt.param_groups[0]["lr"]) < 1e-12


def test_balanced_scene_shards():
    """Data-parallel sharding by crowd size: equal scene counts, disjoint, near-equal summed crowd sizes."""
    import bench
    from social_stgcnn_amd.trainer import shard_scenes
    counts = bench.ragged_counts(2051, seed=5)
    for world in (2, 8):
        shards = [shard_scenes(counts, world, r) for r in range(world)]
        per = len(counts) // world
        assert all(len(s) == per for s in shards)
        allidx = np.concatenate(shards)
        assert len(np.unique(allidx)) == per * world
        sums = np.array([counts[s].sum() for s in shards], dtype=np.float64)
        assert sums.max() / sums.min() < 1.01, sums
        plain = np.array([counts[shard_scenes(counts, world, r, balanced=False)].sum() for r in range(world)], dtype=np.float64)
        assert sums.max() - sums.min() <= plain.max() - plain.min()


def test_bench_rank_launcher_is_supervised():
    """bench.py --gpus N self-launches its ranks as fresh children and supervises them: more RCCL
    ranks than visible GPUs are refused before anything is spawned; a rank that dies (here: no GPU in this container) takes
    the others down and the parent exits non-zero, quickly, with the failing rank's stderr."""
    import subprocess
    import sys
    import time
    if torch.cuda.device_count() > 0:
        pytest.skip("needs a machine without GPUs: the ranks are meant to fail")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"]
    env = {k: v for k, v in os.environ.items() if k != "STG_DIST_BACKEND"}
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "visible GPUs" in (r.stderr + r.stdout)
    t0 = time.time()
    r = subprocess.run(cmd, env=dict(env, STG_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "failed" in r.stderr and "the other ranks were stopped" in r.stderr, r.stderr[-500:]
    assert time.time() - t0 < 200
