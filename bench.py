#!/usr/bin/env python3
"""bench.py -- scene-windows/s, forward + loss + backward + SGD update, of the fused HIP path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic scene-windows already resident in
HBM: model forward (st_gcn + TXP-CNN) -> bivariate NLL -> backward -> [gradient all-reduce over
RCCL when N > 1] -> SGD update.  Workload (BASELINE.json north-star): obs 8 / pred 12, V = 32
pedestrians, 2048 scene-windows per GPU, fp32; synthetic trajectories per SURVEY 8d (random-walk
recipe of complete_nuscenes_setup.py:264-286), random-init weights (torch.manual_seed(0)).

Rank 0 prints ONE JSON line; besides the contract fields it carries
  roofline     the backward entry point (its kernels) algorithmic FLOP/s vs the fp32 MFMA/vector peak,
               timed with HIP events on the launch stream inside the timed region,
  cpu_baseline the CPU oracle run the way the reference runs (one scene per forward, N = 1) on a
               bounded sample of the same workload, rank 0 / N = 1 only,
  kernels      stand-alone adj_build / spatial_agg HBM GB/s (the north-star's bandwidth kernels).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

T_OBS, T_PRED = 8, 12
PEAK_FP32_TFLOPS = 157.3      # MI355X fp32 vector == fp32-input MFMA dense peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def synth_scenes(n, v, seed):
    """start ~ U(-10,10)^2, velocity ~ U(-.5,.5)^2, position noise N(0,.1), velocity noise
    N(0,.05) clipped to +-1, rounded to 4 decimals; rel = first difference with rel[0] = 0."""
    rng = np.random.default_rng(seed)
    t_all = T_OBS + T_PRED
    pos = rng.uniform(-10, 10, (n, v, 2))
    vel = rng.uniform(-0.5, 0.5, (n, v, 2))
    traj = np.zeros((n, v, 2, t_all))
    for t in range(t_all):
        traj[..., t] = pos + rng.normal(0, 0.1, (n, v, 2))
        vel = np.clip(vel + rng.normal(0, 0.05, (n, v, 2)), -1, 1)
        pos = pos + vel
    traj = np.around(traj, 4)
    rel = np.zeros_like(traj)
    rel[..., 1:] = traj[..., 1:] - traj[..., :-1]
    obs_rel = rel[..., :T_OBS].astype(np.float32)                       # (N,V,2,8)
    pred_rel = rel[..., T_OBS:].astype(np.float32)
    target = np.ascontiguousarray(np.transpose(pred_rel, (0, 3, 1, 2)))  # (N,P,V,2)
    return obs_rel, target


# pedestrians per scene-window of eth/train (2785 windows, mean 10.7, median 5, max 57), index = V:
# measured with social_stgcnn_amd.data.load_windows on the dataset's text files -- the shape of real ragged batches
ETH_TRAIN_PEDS_HIST = [0, 0, 559, 371, 319, 226, 163, 117, 84, 84, 46, 27, 24, 31, 16, 19, 17, 22, 32, 23, 31, 35, 40,
                       36, 35, 37, 22, 16, 19, 28, 29, 24, 31, 29, 30, 16, 20, 20, 21, 22, 7, 5, 6, 9, 4, 7, 6, 2, 4, 3,
                       1, 5, 2, 2, 0, 0, 0, 1]


def ragged_counts(n, seed, order="shuffled"):
    """n pedestrian counts drawn from the eth/train histogram (BASELINE configs[1]: 'ETH train, batch=512')."""
    h = np.asarray(ETH_TRAIN_PEDS_HIST, dtype=np.float64)
    c = np.random.default_rng(seed).choice(len(h), size=n, p=h / h.sum()).astype(np.int32)
    if order == "sorted":
        c = np.sort(c)[::-1].copy()
    return c


def flops_per_window(v, fwd=True, bwd=True):
    """SURVEY 8d algorithmic work per scene-window (matches torch FlopCounterMode on the reference)."""
    f = 62000 * v + 80 * v * v
    tot = 185680 * v + 160 * v * v
    return (f if fwd else 0) + ((tot - f) if bwd else 0)


def bytes_per_window(v):
    return 608 * v + 64 * v * v


def time_kernel(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def cpu_baseline(obs_rel, target, budget_s=15.0, counts=None):
    """The oracle, driven like train.py:36-77 drives the reference: one scene per forward (N = 1),
    loss, backward -- single thread, bounded sample of the same workload."""
    from oracle import stgcnn_oracle as O
    torch.set_num_threads(1)
    torch.manual_seed(0)
    from social_stgcnn_amd.model import social_stgcnn
    m = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12)
    state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    keys = [k for k, _ in m.named_parameters()]
    scenes = []
    for i in range(min(64, obs_rel.shape[0])):
        c = obs_rel.shape[1] if counts is None else int(counts[i])
        nodes, lap = O.seq_to_graph_np(obs_rel[i, :c])
        scenes.append((torch.from_numpy(nodes).unsqueeze(0).permute(0, 3, 1, 2), torch.from_numpy(lap),
                       torch.from_numpy(np.ascontiguousarray(target[i, :, :c]))))
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        x, a, tgt = scenes[done % len(scenes)]
        params = {k: state[k].clone().requires_grad_(True) for k in keys}
        work = dict(state)
        work.update(params)
        loss, _ = O.scene_loss(work, x, a, tgt, True)
        loss.backward()
        done += 1
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "scene-windows/s", "cores": 1, "kind": "port",
            "sample": "%d scene-windows (V=%d, N=1 per forward like train.py:36-77) in %.1f s, oracle on torch CPU "
                      "ops, 1 thread" % (done, obs_rel.shape[1], dt)}


def pmc_traffic(v, n):
    """HBM bytes per stg_model_bwd launch from the committed rocprofv3 PMC passes (tools/gpu_traffic.sh:
    FETCH_SIZE and WRITE_SIZE in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950).  Counters cannot be collected inside this process, so the figure is the profiled one and only
    reported for the workload it was profiled on (V=32, 2048 scene-windows); otherwise null."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
    if (v, n) != (32, 2048) or not os.path.exists(path):
        return None
    with open(path) as f:
        t = json.load(f)
    ks = ("txp_bwd_wave_kernel", "model_bwd_kernel", "txp_wgrad_kernel", "reduce_slabs_kernel")
    if not all(k in t and "hbm_bytes_per_launch" in t[k] for k in ks):
        return None
    return sum(t[k]["hbm_bytes_per_launch"] for k in ks)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=2048, help="scene-windows per GPU")
    ap.add_argument("--peds", type=int, default=32, help="pedestrians per scene-window (V)")
    ap.add_argument("--ragged", choices=("shuffled", "sorted"), default=None,
                    help="ragged batch: pedestrians per scene drawn from the eth/train histogram (padded to the "
                         "largest draw); 'sorted' orders the scenes by crowd size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL ("nccl") over xGMI on a real node; STG_DIST_BACKEND=gloo lets several ranks share ONE GPU to
        # rehearse the multi-rank code path on a single-GPU box
        backend = os.environ.get("STG_DIST_BACKEND", "nccl")
        local = local % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from social_stgcnn_amd import ops
    from social_stgcnn_amd.model import social_stgcnn
    from social_stgcnn_amd.trainer import Trainer, broadcast_module

    n, v = args.batch, args.peds
    counts, peds_d = None, None
    if args.ragged:
        counts = ragged_counts(n, seed=1 + rank, order=args.ragged)
        v = int(counts.max())
    obs_rel, target = synth_scenes(n, v, seed=1 + rank)
    if counts is not None:
        live = np.arange(v)[None, :] < counts[:, None]                  # (N,V)
        obs_rel *= live[:, :, None, None]
        target *= live[:, None, :, None]
        peds_d = torch.from_numpy(counts).to(dev)
    rel_d = torch.from_numpy(obs_rel).to(dev)
    tgt_d = torch.from_numpy(target).to(dev)
    nodes, adj = ops.adj_build(rel_d, peds_d)             # graph build stays on the device
    x = nodes.permute(0, 3, 1, 2)                         # (N,2,T,V) strided view like train.py:48
    weights = torch.full((n,), 1.0 / (n * world), device=dev)

    torch.manual_seed(0)
    model = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=T_OBS, kernel_size=3,
                          pred_seq_len=T_PRED).to(dev).train()
    broadcast_module(model)
    trainer = Trainer(model, lr=0.01)

    if args.no_graph:
        def step():
            return trainer.step(x, adj, tgt_d, peds_d, weights)
    else:
        step = trainer.capture(x, adj, tgt_d, peds_d, weights)      # the whole step as one hipGraph
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ops.TIMER = ops.KernelTimer()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # per-call device time of the forward / backward entry points: the same K steps once more, launched
    # eagerly with HIP-event brackets on the launch stream (a replayed hipGraph cannot be bracketed per node)
    for _ in range(args.steps):
        trainer.step(x, adj, tgt_d, peds_d, weights)
    torch.cuda.synchronize()
    timer, ops.TIMER = ops.TIMER, None
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        value = world * n * args.steps / elapsed
        bwd_ms = timer.mean_ms("model_bwd")
        fwd_ms = timer.mean_ms("model_fwd")
        per_scene = [v] * n if counts is None else [int(c) for c in counts]
        bwd_flops = sum(flops_per_window(c, fwd=False) for c in per_scene)
        fwd_flops = sum(flops_per_window(c, bwd=False) for c in per_scene)
        all_flops = sum(flops_per_window(c) for c in per_scene) / n
        all_bytes = sum(bytes_per_window(c) for c in per_scene) / n
        if counts is None:
            workload = ("synthetic V=%d scene-windows, obs 8 / pred 12, batch %d per GPU, fp32 (BASELINE north-star: "
                        "V<=32, batch=2048; SURVEY 8d generator)" % (v, n))
        else:
            workload = ("synthetic ragged scene-windows, pedestrians per window drawn from the eth/train histogram "
                        "(mean %.1f, max %d, %s order), obs 8 / pred 12, batch %d per GPU, fp32 (BASELINE configs[1] "
                        "shape)" % (float(counts.mean()), v, args.ragged, n))
        achieved = bwd_flops / (bwd_ms * 1e-3) / 1e12
        out = {
            "metric": "scene-windows/sec fwd+bwd (obs=8,pred=12)",
            "value": value, "unit": "scene-windows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "launch": "eager" if args.no_graph else "hipGraph replay",
            "config": {"workload": workload,
                       "global_batch": n * world, "step": "forward + bivariate NLL + backward + "
                       + ("RCCL all-reduce + " if world > 1 else "") + "SGD update",
                       "parallelism": "dp%d" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_TFLOPS, "traffic": pmc_traffic(v, n) if counts is None else None,
                         "traffic_note": "HBM bytes per stg_model_bwd launch, profiles/r01_pmc_traffic.json "
                                         "(rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH x2)",
                         "kernel": "stg_model_bwd = txp_bwd_wave_kernel + model_bwd_kernel + txp_wgrad_kernel + "
                                   "reduce_slabs_kernel (the backward of one batch)", "launch_ms": bwd_ms,
                         "algorithmic_flop_per_launch": bwd_flops,
                         "fwd_kernel": {"kernel": "stg_model_fwd = model_fwd_kernel + txp_fwd_wave_kernel", "launch_ms": fwd_ms,
                                        "achieved": fwd_flops / (fwd_ms * 1e-3) / 1e12}},
            "end_to_end": {"algorithmic_tflops": all_flops * value / 1e12,
                           "algorithmic_gbs": all_bytes * value / 1e9},
        }
        # stand-alone bandwidth kernels (north-star: achieved HBM GB/s vs the gfx950 peak)
        adj_ms = time_kernel(lambda: ops.adj_build(rel_d, peds_d))
        agg_in = torch.randn(n, 5, T_OBS, v, device=dev)
        agg_ms = time_kernel(lambda: ops.spatial_agg(agg_in, adj))
        adj_bytes = n * (64 * v + 32 * v * v + 64 * v)
        agg_bytes = n * (32 * v * v + 2 * 160 * v)
        out["kernels"] = {
            "adj_build": {"ms": adj_ms, "GBps": adj_bytes / adj_ms / 1e6, "frac_hbm": adj_bytes / adj_ms / 1e6 / PEAK_HBM_GBS},
            "spatial_agg_fwd": {"ms": agg_ms, "GBps": agg_bytes / agg_ms / 1e6,
                                "frac_hbm": agg_bytes / agg_ms / 1e6 / PEAK_HBM_GBS},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(obs_rel, target, args.cpu_seconds, counts)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
