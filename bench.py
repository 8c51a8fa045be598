#!/usr/bin/env python3
"""bench.py -- scene-windows/s, forward + loss + backward + SGD update, of the fused HIP path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` without a launcher (no WORLD_SIZE in the environment) starts the N ranks itself, as child processes,
before anything touches the GPU; rank 0's JSON line is the output and a failing rank fails the run.

A "step" is one pass of the hot path over one batch of scene-windows already resident in HBM: adjacency-weighted
aggregation -> st_gcn block + TXP-CNN -> bivariate NLL -> backward -> [ONE all-reduce over RCCL when N > 1] -> SGD
update, replayed as a hipGraph.  Workload (BASELINE.json north-star): obs 8 / pred 12, V = 32 pedestrians, 2048
scene-windows per GPU, fp32; synthetic trajectories per SURVEY 8d (random-walk recipe of
complete_nuscenes_setup.py:264-286), random-init weights (torch.manual_seed(0)).  `--dataset eth-train --batch 512` is
BASELINE configs[1] on the real (ragged) eth/train windows that travel with the repo as a test fixture.

Timing: W warm-up steps, then R blocks of EXACTLY K steps, each block bracketed by barrier + torch.cuda.synchronize()
on both sides (MAX over ranks); `ms_per_step` / `value` are the MEDIAN block, p10 / p90 ride along (R is chosen so the
timed region lasts >= 1 s).  Two different input batches alternate from step to step.

Rank 0 prints ONE JSON line; besides the contract fields it carries
  roofline     the DOMINANT SINGLE KERNEL of the step: algorithmic FLOP per launch / its mean duration, measured live
               with HIP events the library records between its kernels on the launch stream (`composite` = the whole
               backward entry point as last round),
  pipeline     the same step with the adjacency build (utils.seq_to_graph as a HIP kernel) inside the graph,
  kernels      stand-alone HBM kernels on a working set beyond the 256 MiB Infinity Cache,
  cpu_baseline the CPU oracle run the way the reference runs (one scene per forward, N = 1) on a bounded sample of the
               same workload: 1 thread, plus `cpu_baseline_all_cores` (one single-threaded worker per core).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

T_OBS, T_PRED = 8, 12
PEAK_FP32_TFLOPS = 157.3      # MI355X fp32 vector == fp32-input MFMA dense peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=0, help="timed blocks of --steps steps (0: >= 30 and >= 1 s)")
    ap.add_argument("--batch", type=int, default=2048, help="scene-windows per GPU")
    ap.add_argument("--peds", type=int, default=32, help="pedestrians per scene-window (V)")
    ap.add_argument("--ragged", choices=("shuffled", "sorted"), default=None,
                    help="ragged batch: pedestrians per scene drawn from the eth/train histogram (padded to the "
                         "largest draw); 'sorted' orders the scenes by crowd size")
    ap.add_argument("--dataset", choices=("synthetic", "eth-train", "all-train"), default="synthetic",
                    help="eth-train: real ragged windows of tests/golden/data/eth_train (BASELINE configs[1])")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="bf16: bf16 STORAGE of saved activations / hand-offs (fp32 accumulate, fp32 parameters)")
    ap.add_argument("--wg-path", action="store_true", help="force the workgroup-per-scene kernels (STG_OPT_WG_PATH)")
    ap.add_argument("--wave-path", action="store_true",
                    help="keep the wave-per-scene kernels for a small batch too (STG_OPT_WAVE_PATH)")
    ap.add_argument("--f32-mfma", action="store_true",
                    help="A/B: the fp32-MFMA convolution / weight-gradient kernels instead of the exact bf16-pipe ones "
                         "(STG_OPT_F32_MFMA)")
    ap.add_argument("--wg-waves", type=int, default=0, help="waves per scene of the workgroup-per-scene kernels (0 auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-extras", action="store_true", help="skip the pipeline / stand-alone kernel / roofline legs")
    ap.add_argument("--kernels-only", action="store_true",
                    help="of the extra legs keep only the per-kernel device times (roofline); no pipeline / epoch / HBM kernels")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="seconds the self-launched ranks of --gpus N may take before the parent stops them")
    ap.add_argument("--cpu-worker", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--probe-fds", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------
# self-launch: N ranks as child processes, before this process touches the GPU
# ------------------------------------------------------------------------------------------
def launch_ranks(args):
    """N ranks as FRESH child processes of a parent that never touches the GPU.  The children are supervised: the first
    one that exits non-zero takes the others down (SIGTERM, then SIGKILL) and the parent exits non-zero with the tail of
    the failing rank's stderr -- a rank that dies during RCCL initialisation must not leave its peers waiting in
    init_process_group until the caller's time limit."""
    import socket
    import tempfile
    import torch
    backend = os.environ.get("STG_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()                     # (counts devices without initialising the GPU runtime)
    if backend == "nccl" and args.gpus > n_dev:
        raise SystemExit("bench.py: --gpus %d over RCCL needs %d visible GPUs, found %d (STG_DIST_BACKEND=gloo rehearses "
                         "several ranks on one GPU)" % (args.gpus, args.gpus, n_dev))
    with socket.socket() as s:                            # (a free port now; the ranks bind it a moment later)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, errs = [], []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        err = tempfile.TemporaryFile(mode="w+")
        errs.append(err)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=err, text=True))
    out0 = []
    import threading
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()                                        # (rank 0's pipe is drained while the parent polls)
    deadline = time.monotonic() + args.launch_timeout
    failed = None
    while failed is None:
        codes = [p.poll() for p in procs]
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = (bad[0], "exit code %d" % codes[bad[0]])
        elif all(c == 0 for c in codes):
            break
        elif time.monotonic() > deadline:
            failed = (next(r for r, c in enumerate(codes) if c is None), "still running after %.0f s" % args.launch_timeout)
        else:
            time.sleep(0.2)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_kill = time.monotonic() + 10
        for p in procs:
            try:
                p.wait(max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
        r, why = failed
        errs[r].seek(0)
        tail = errs[r].read()[-2000:]
        sys.stderr.write("bench.py: rank %d of %d failed (%s); the other ranks were stopped.\n--- rank %d stderr (tail) ---\n%s\n"
                         % (r, args.gpus, why, r, tail))
        raise SystemExit(1)
    reader.join(10)
    sys.stdout.write(out0[0] if out0 else "")
    return 0


# ------------------------------------------------------------------------------------------
# workload
# ------------------------------------------------------------------------------------------
def synth_scenes(n, v, seed):
    """start ~ U(-10,10)^2, velocity ~ U(-.5,.5)^2, position noise N(0,.1), velocity noise
    N(0,.05) clipped to +-1, rounded to 4 decimals; rel = first difference with rel[0] = 0."""
    import numpy as np
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
    """n pedestrian counts drawn from the eth/train histogram."""
    import numpy as np
    h = np.asarray(ETH_TRAIN_PEDS_HIST, dtype=np.float64)
    c = np.random.default_rng(seed).choice(len(h), size=n, p=h / h.sum()).astype(np.int32)
    if order == "sorted":
        c = np.sort(c)[::-1].copy()
    return c


def real_windows(which):
    """the scene-windows of a real dataset: "eth-train" = the reference's datasets/eth/train (2,785 windows; BASELINE
    configs[1]); "all-train" = the five leave-one-out ETH/UCY train sets concatenated (11,889 windows; BASELINE configs[2]),
    built from the eight recordings committed under tests/golden/data."""
    from social_stgcnn_amd import data
    gd = os.path.join(ROOT, "tests", "golden", "data")
    if which == "eth-train":
        return data.load_windows(os.path.join(gd, "eth_train"), T_OBS, T_PRED, 1, with_non_linear=False)
    splits = data.load_train_splits([os.path.join(gd, "eth_train"), os.path.join(gd, "train_extra")], obs_len=T_OBS,
                                    pred_len=T_PRED)
    return data.concat_windows([splits[k] for k in ("eth", "hotel", "univ", "zara1", "zara2")])


def eth_train_batches(n, n_batches, seed, which="eth-train"):
    """n_batches batches of n real windows (seeded shuffle of the dataset, like the reference's DataLoader(shuffle=True)),
    each padded to ITS largest crowd (rounded up to 4): (obs_rel (N,V,2,8), target (N,P,V,2), counts)."""
    import numpy as np
    from social_stgcnn_amd import data
    win = real_windows(which)
    perm = np.random.default_rng(seed).permutation(len(win))
    out = []
    for b in range(n_batches):
        idx = np.sort(perm[(b * n) % len(win):][:n]) if (b * n) % len(win) + n <= len(win) else np.sort(perm[:n])
        v_pad = (int(win.num_peds[idx].max()) + 3) & ~3          # (16-byte aligned adjacency rows)
        obs_rel, pred_rel, _, _, counts = data.pad_batch(win, idx, v_pad=v_pad)
        out.append((np.ascontiguousarray(np.transpose(obs_rel, (0, 2, 3, 1))), pred_rel, counts))
    return out, len(win)


def flops_per_window(v, fwd=True, bwd=True):
    """SURVEY 8d algorithmic work per scene-window (matches torch FlopCounterMode on the reference)."""
    f = 62000 * v + 80 * v * v
    tot = 185680 * v + 160 * v * v
    return (f if fwd else 0) + ((tot - f) if bwd else 0)


def bytes_per_window(v):
    return 608 * v + 64 * v * v


# algorithmic FLOP per scene-window of the kernels behind the fused entry points, in launch order (SURVEY 8d split:
# TXP convs 60,480 V forward, the same again for the input gradients and for the weight gradients; gcn/tcn/residual
# convs 1,520 V; the aggregation 80 V^2 with its 2x re-association saving not credited)
KERNELS = {
    "model_fwd": [("stgcn_agg_kernel (x A, colsum A: the read of A)", lambda v: 80 * v * v, "hbm"),
                  ("txp_fwd_x6_kernel / txp_fwd_wave_kernel (st_gcn block + TXP-CNN forward)", lambda v: 62000 * v, "mfma")],
    "model_bwd": [("txp_bwd_x6_kernel / txp_bwd_wave_kernel (loss gradient + TXP input-gradient chain + st_gcn block "
                   "backward)", lambda v: 60480 * v + 2 * 1520 * v + 80 * v * v, "mfma"),
                  ("txp_wgrad_bf16_kernel / txp_wgrad_kernel (TXP weight / bias gradients)", lambda v: 60480 * v, "mfma"),
                  ("reduce_slabs_kernel (+ SGD, BatchNorm fold, reported loss)", lambda v: 0, "hbm")],
}
PEAK_BF16_TFLOPS = 2516.6     # dense v_mfma_f32_16x16x32_bf16: 256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz


def issued_bf16_flop(kernel, v):
    """FLOP the exact-bf16 kernels ISSUE per scene-window on v_mfma_f32_16x16x32_bf16 (16,384 FLOP each): every fp32
    product is six bf16 products, M = 12 of 16 rows and K = 108 of 128 are real.  None for the fp32-MFMA kernels
    (V > 32, bf16 storage: the library then runs txp_bwd_wave_kernel / txp_wgrad_kernel)."""
    if v > 32:
        return None
    if kernel.startswith("txp_fwd_x6") or kernel.startswith("txp_bwd_x6"):
        return 5 * ((5 * v + 15) // 16) * 24 * 16384          # five convs (n_txpcnn = 5: tpcnns.4 is dead), 24 MFMAs per tile
    if kernel.startswith("txp_wgrad_bf16"):
        return 5 * ((5 * v + 31) // 32) * 57 * 16384          # five layers, 9 taps x 6 products + 3 bias MFMAs per 32 positions
    return None


def time_kernel(torch, fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


# ------------------------------------------------------------------------------------------
# CPU baseline: the oracle, driven like train.py:36-77 drives the reference
# ------------------------------------------------------------------------------------------
def cpu_worker_loop(v, budget_s, seed, counts=None):
    """one single-threaded worker: oracle forward + loss + backward, one scene per forward, for budget_s seconds"""
    import numpy as np
    import torch
    from oracle import stgcnn_oracle as O
    from social_stgcnn_amd.model import social_stgcnn
    torch.set_num_threads(1)
    torch.manual_seed(0)
    obs_rel, target = synth_scenes(32, v, seed)
    m = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12)
    state = {k: t.detach().clone() for k, t in m.state_dict().items()}
    keys = [k for k, _ in m.named_parameters()]
    scenes = []
    for i in range(obs_rel.shape[0]):
        c = v if counts is None else max(1, int(counts[i % len(counts)]))
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
    return done, time.perf_counter() - t0


def kernel_source_hash():
    """sha1 over the kernel sources (csrc/*.hip, *.hpp, include/*.h): what a committed PMC traffic file was measured on
    (tools/profile_all.sh stamps it); bench.py quotes the file only for the same sources."""
    import glob
    import hashlib
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(ROOT, "social_stgcnn_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "social_stgcnn_amd", "csrc", "*.hpp")) +
                   glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def host_cores():
    """cores this job may use: the affinity mask, cut down to the cgroup's CPU quota where there is one"""
    n = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baselines(v, budget_s, counts=None):
    done, dt = cpu_worker_loop(v, budget_s, 1, counts)
    one = {"value": done / dt, "unit": "scene-windows/s", "cores": 1, "kind": "port",
           "sample": "%d scene-windows (V=%s, N=1 per forward like train.py:36-77) in %.1f s, oracle on torch CPU ops, "
                     "1 thread" % (done, v if counts is None else "ragged", dt)}
    avail = host_cores()
    # the workers are fresh children that never need the GPU: it is HIDDEN from them (torch's autograd engine would
    # otherwise open the device node even for CPU work, and the GPU box admits only a few processes holding it), so
    # "all cores" means all cores the box gives this job (they share nothing and scale linearly)
    hidden = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="",
                  CUDA_VISIBLE_DEVICES="")
    # (the visibility variables alone do not keep torch's ROCm runtime from opening the device node: oracle/nogpu_shim.c,
    # preloaded into the workers only, makes the node look absent.  A probe worker checks that it holds no device node
    # before all the workers start; otherwise stay inside the box's bound of processes holding the GPU.)
    shim = os.path.join(ROOT, "oracle", "libnogpu_shim.so")
    clean = False
    if os.path.exists(shim):
        hidden["LD_PRELOAD"] = (shim + " " + os.environ.get("LD_PRELOAD", "")).strip()
        try:
            probe = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", "0", "--peds", "2",
                                    "--cpu-seconds", "0.2", "--probe-fds"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                   text=True, env=hidden, timeout=300)
            clean = probe.returncode == 0 and probe.stdout.strip().endswith("no-device-node")
        except (OSError, subprocess.SubprocessError):
            clean = False
    cores = min(avail, 64) if clean else min(avail, 4)
    if not clean:
        hidden.pop("LD_PRELOAD", None)
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(i), "--peds", str(v),
                               "--cpu-seconds", str(budget_s)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                              text=True, env=hidden)
             for i in range(cores)]
    tot, tmax = 0, 0.0
    for p in procs:
        o, _ = p.communicate()
        try:
            d, t = o.strip().split()[-2:]
            tot += int(d)
            tmax = max(tmax, float(t))
        except Exception:
            pass
    many = {"value": tot / tmax if tmax > 0 else None, "unit": "scene-windows/s", "cores": cores, "kind": "port",
            "host_cores_available": avail,
            "sample": "%d scene-windows by %d single-threaded worker processes%s (data parallel over scenes like the "
                      "GPU path; %d cores available to this job) in %.1f s"
                      % (tot, cores, " with the GPU hidden from them" if clean else
                         ", capped at 4: they could not be kept from opening the GPU device node", avail, tmax)}
    return one, many


# ------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    if args.cpu_worker >= 0:                  # child of cpu_baselines(): never touches the GPU
        d, t = cpu_worker_loop(args.peds, args.cpu_seconds, 100 + args.cpu_worker)
        print(d, t)
        if args.probe_fds:                    # does this process hold a GPU device node after a backward pass?
            held = []
            for fd in os.listdir("/proc/self/fd"):
                try:
                    tgt = os.readlink("/proc/self/fd/" + fd)
                except OSError:
                    continue
                if tgt == "/dev/kfd" or tgt.startswith("/dev/dri/"):
                    held.append(tgt)
            print("holds %s" % held if held else "no-device-node")
        return 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)             # (before any torch.cuda call in this process)
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL ("nccl") over xGMI on a real node; STG_DIST_BACKEND=gloo lets several ranks share ONE GPU to rehearse
        # the multi-rank code path on a single-GPU box
        backend = os.environ.get("STG_DIST_BACKEND", "nccl")
        n_dev = max(1, torch.cuda.device_count())
        if backend == "nccl" and world > n_dev:
            raise SystemExit("bench.py: %d RCCL ranks need %d visible GPUs, found %d" % (world, world, n_dev))
        local = local % n_dev                      # (several gloo ranks may share one GPU: the one-GPU rehearsal)
        torch.cuda.set_device(local)
        import datetime
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from social_stgcnn_amd import ops
    from social_stgcnn_amd.model import social_stgcnn
    from social_stgcnn_amd.trainer import Trainer, broadcast_module
    options = ops.KernelOptions(wg_path=bool(args.wg_path), f32_mfma=bool(args.f32_mfma), wave_path=bool(args.wave_path),
                                wg_waves=int(args.wg_waves), bf16_store=args.dtype == "bf16")

    n, v = args.batch, args.peds
    n_sets = 2                                  # input batches that alternate from step to step
    sets, n_windows = [], None
    real = args.dataset in ("eth-train", "all-train")
    if real:
        raw, n_windows = eth_train_batches(n, n_sets, seed=1 + rank, which=args.dataset)
        for obs_rel, target, counts in raw:
            sets.append((obs_rel, target, counts))
    else:
        for k in range(n_sets):
            counts = None
            vv = v
            if args.ragged:
                counts = ragged_counts(n, seed=1 + rank + 101 * k, order=args.ragged)
                vv = int(counts.max())
            obs_rel, target = synth_scenes(n, vv, seed=1 + rank + 101 * k)
            if counts is not None:
                live = np.arange(vv)[None, :] < counts[:, None]
                obs_rel *= live[:, :, None, None]
                target *= live[:, None, :, None]
            sets.append((obs_rel, target, counts))
    dsets = []
    for obs_rel, target, counts in sets:
        rel_d = torch.from_numpy(obs_rel).to(dev)
        tgt_d = torch.from_numpy(target).to(dev)
        peds_d = None if counts is None else torch.from_numpy(counts.astype(np.int32)).to(dev)
        nodes, adj = ops.adj_build(rel_d, peds_d)             # graph build stays on the device
        dsets.append(dict(rel=rel_d, tgt=tgt_d, peds=peds_d, nodes=nodes, adj=adj,
                          x=nodes.permute(0, 3, 1, 2), v=obs_rel.shape[1]))   # (N,2,T,V) strided view like train.py:48
    weights = torch.full((n,), 1.0 / (n * world), device=dev)

    torch.manual_seed(0)
    model = social_stgcnn(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=T_OBS, kernel_size=3,
                          pred_seq_len=T_PRED).to(dev).train()
    model.options = options
    broadcast_module(model)
    init_state = {k: t.detach().clone() for k, t in model.state_dict().items()}
    trainer = Trainer(model, lr=0.01)

    def eager(d):
        return lambda: trainer.step(d["x"], d["adj"], d["tgt"], d["peds"], weights)
    if args.no_graph:
        steps = [eager(d) for d in dsets]
    else:
        steps = [trainer.capture(d["x"], d["adj"], d["tgt"], d["peds"], weights) for d in dsets]

    def run_block(fns, k):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for i in range(k):
            fns[i % len(fns)]()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    for i in range(args.warmup):
        steps[i % len(steps)]()
    first = run_block(steps, args.steps)
    repeats = args.repeats if args.repeats > 0 else max(30, int(1.0 / max(first, 1e-6)) + 1)
    if world > 1:                                   # every rank must run the same number of blocks
        rr = torch.tensor([repeats], device=dev, dtype=torch.int64)
        dist.all_reduce(rr, op=dist.ReduceOp.MAX)
        repeats = int(rr.item())
    blocks = [first] + [run_block(steps, args.steps) for _ in range(repeats - 1)]
    per_step = np.sort(np.asarray(blocks) / args.steps * 1e3)
    ms_step = float(np.median(per_step))
    p10, p90 = float(np.percentile(per_step, 10)), float(np.percentile(per_step, 90))

    out = None
    if rank == 0:
        value = world * n * 1e3 / ms_step
        per_scene = []
        for d, (_, _, counts) in zip(dsets, sets):
            per_scene += [d["v"]] * n if counts is None else [int(c) for c in counts]
        all_flops = sum(flops_per_window(c) for c in per_scene) / len(per_scene)
        all_bytes = sum(bytes_per_window(c) for c in per_scene) / len(per_scene)
        if real:
            what = ("eth/train scene-windows (BASELINE configs[1]" if args.dataset == "eth-train" else
                    "ETH/UCY scene-windows, the five leave-one-out train sets concatenated (BASELINE configs[2]")
            workload = ("REAL %s: %d of the %d windows per batch, seeded shuffle, 2..57 pedestrians, mean %.1f, padded to "
                        "%d), obs 8 / pred 12, batch %d per GPU, %s"
                        % (what, n, n_windows, float(np.mean(per_scene)), max(d["v"] for d in dsets), n, args.dtype))
        elif args.ragged:
            workload = ("synthetic ragged scene-windows, pedestrians per window drawn from the eth/train histogram "
                        "(mean %.1f, max %d, %s order), obs 8 / pred 12, batch %d per GPU, %s"
                        % (float(np.mean(per_scene)), max(d["v"] for d in dsets), args.ragged, n, args.dtype))
        else:
            workload = ("synthetic V=%d scene-windows, obs 8 / pred 12, batch %d per GPU, %s (BASELINE north-star: "
                        "V<=32, batch=2048; SURVEY 8d generator)" % (v, n, args.dtype))
        out = {
            "metric": "scene-windows/sec fwd+bwd (obs=8,pred=12)",
            "value": value, "unit": "scene-windows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.dtype == "f32" else "bf16-storage (fp32 accumulate, fp32 parameters)",
            "data": ("real (%s)" % args.dataset) if real else "synthetic",
            "launch": "eager" if args.no_graph else "hipGraph replay",
            "timing": {"blocks": len(blocks), "steps_per_block": args.steps, "ms_per_step_median": ms_step,
                       "ms_per_step_p10": p10, "ms_per_step_p90": p90, "ms_per_step_first_block": first / args.steps * 1e3,
                       "timed_region_s": float(np.sum(blocks)), "input_batches_rotated": len(steps)},
            "config": {"workload": workload, "global_batch": n * world,
                       "step": "aggregation + forward + bivariate NLL + backward + "
                               + ("ONE RCCL all-reduce + " if world > 1 else "") + "SGD update",
                       "parallelism": "dp%d" % world},
            "end_to_end": {"algorithmic_tflops": all_flops * value / 1e12, "algorithmic_gbs": all_bytes * value / 1e9},
        }

    # ---- per-kernel device time (HIP events the library records between its kernels) -> roofline ----------------
    if not args.no_extras:
        model.timer = ops.KernelTimer()
        for i in range(max(4, min(args.steps, 20))):
            eager(dsets[i % len(dsets)])()
        torch.cuda.synchronize()
        timer, model.timer = model.timer, None
        if rank == 0:
            vmean = float(np.mean([c * 1.0 for c in per_scene]))
            kern = []
            for entry in ("model_fwd", "model_bwd"):
                ms = timer.kernel_ms(entry)
                for (name, flop, bound), t in zip(KERNELS[entry], ms):
                    fl = sum(flop(c) for c in per_scene) / len(dsets)
                    row = {"kernel": name, "entry": "stg_" + entry, "launch_ms": t, "bound": bound,
                           "algorithmic_flop_per_launch": fl,
                           "achieved_tflops": fl / (t * 1e-3) / 1e12 if t > 0 else None}
                    iss = ([issued_bf16_flop(name, c) for c in per_scene]
                           if args.dtype == "f32" and not args.f32_mfma and not args.wg_path else [None])
                    if t > 0 and all(q is not None for q in iss):
                        ifl = sum(iss) / len(dsets)
                        row["issued"] = {"instruction": "v_mfma_f32_16x16x32_bf16 (exact three-piece operands: six bf16 "
                                                        "products per fp32 product)",
                                         "flop_per_launch": ifl, "tflops": ifl / (t * 1e-3) / 1e12,
                                         "peak": PEAK_BF16_TFLOPS, "frac": ifl / (t * 1e-3) / 1e12 / PEAK_BF16_TFLOPS}
                    kern.append(row)
            dom = max((k for k in kern if k["bound"] == "mfma"), key=lambda k: k["launch_ms"])
            bwd_ms = sum(k["launch_ms"] for k in kern if k["entry"] == "stg_model_bwd")
            bwd_fl = sum(flops_per_window(c, fwd=False) for c in per_scene) / len(dsets)
            traffic, traffic_src = None, "no PMC file for this configuration"
            tpath = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
            if (os.path.exists(tpath) and args.dataset == "synthetic" and not args.ragged and (v, n) == (32, 2048)
                    and args.dtype == "f32" and not args.f32_mfma and not args.wg_path and not real):
                with open(tpath) as f:
                    tj = json.load(f)
                key = dom["kernel"].split(" ")[0]
                want = tj.get("_meta", {}).get("kernel_source_sha1")
                if want == kernel_source_hash():
                    traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
                    traffic_src = "profiles/r03_pmc_traffic.json, measured on these kernel sources (sha1 %s)" % want[:12]
                else:
                    traffic_src = ("profiles/r03_pmc_traffic.json was measured on other kernel sources (sha1 %s): not "
                                   "quoted; regenerate with tools/profile_all.sh" % str(want)[:12])
            out["roofline"] = {
                "bound": "mfma", "achieved": dom["achieved_tflops"], "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": dom["achieved_tflops"] / PEAK_FP32_TFLOPS, "traffic": traffic,
                "traffic_note": "HBM bytes per launch of this kernel (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate "
                                "passes, FETCH x2): " + traffic_src,
                "kernel": dom["kernel"], "launch_ms": dom["launch_ms"],
                "algorithmic_flop_per_launch": dom["algorithmic_flop_per_launch"],
                "issued": dom.get("issued"),
                "note": "dominant single MFMA kernel of the step; duration = mean over %d eager launches of HIP-event "
                        "intervals recorded by the library on the launch stream; achieved / peak / frac = ALGORITHMIC "
                        "fp32 FLOP against the fp32-input MFMA peak at 2.4 GHz; `issued` = what the kernel issues on the "
                        "bf16 matrix pipe against that instruction's dense peak"
                        % len(timer.calls["model_bwd"]),
                "kernels": kern,
                "composite": {"kernel": "stg_model_bwd (all its kernels)", "launch_ms": bwd_ms,
                              "algorithmic_flop_per_launch": bwd_fl,
                              "achieved": bwd_fl / (bwd_ms * 1e-3) / 1e12 if bwd_ms > 0 else None,
                              "frac": bwd_fl / (bwd_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS if bwd_ms > 0 else None},
            }
            _ = vmean

        # ---- pipeline: adjacency build inside the step ---------------------------------------------------------
        if not args.no_graph and not args.kernels_only:
            psteps = [trainer.capture(d["x"], d["adj"], d["tgt"], d["peds"], weights,
                                      pre=(lambda d=d: ops.adj_build(d["rel"], d["peds"], out=(d["nodes"], d["adj"]))))
                      for d in dsets]
            for i in range(4):
                psteps[i % len(psteps)]()
            pb = np.sort([run_block(psteps, args.steps) for _ in range(max(5, repeats // 4))]) / args.steps * 1e3
            if rank == 0:
                out["pipeline"] = {"what": "the same step with utils.seq_to_graph (adj_build kernel) inside the captured "
                                           "graph: relative trajectories in, updated weights out",
                                   "ms_per_step": float(np.median(pb)), "value": world * n * 1e3 / float(np.median(pb)),
                                   "unit": "scene-windows/s"}

    # ---- real-data epochs from the device-resident dataset (N1): gather by device index inside the captured step ----
    if real and not args.no_extras and not args.kernels_only and not args.no_graph and world == 1:
        from social_stgcnn_amd.dataset import DeviceWindows, EpochRunner
        win = real_windows(args.dataset)
        ds = DeviceWindows(win, dev)
        # from the seeded initial weights again: thousands of timed steps on two fixed batches leave an over-fitted model
        # whose correlation output can saturate on unseen windows (rho = +-1 -> a NaN loss, like the reference's)
        model.load_state_dict(init_state)
        runner = EpochRunner(trainer, ds, n)
        gen = torch.Generator(device=dev).manual_seed(0)
        runner.train_epoch(ds.shuffled_order(gen))                       # capture + warm-up epoch
        torch.cuda.synchronize()
        n_ep = 20
        ep_losses = []
        t0 = time.perf_counter()
        for _ in range(n_ep):
            last = runner.train_epoch(ds.shuffled_order(gen))
            ep_losses.append(last)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["epoch"] = {"what": "reference-style training epochs (train.py:28-79 group semantics, batch_size %d) over the "
                                "%d real windows resident in HBM: device shuffle, gather by device index, "
                                "adjacency build, step -- ONE captured hipGraph per group, no host->device traffic in "
                                "the loop" % (n, len(ds)),
                        "epochs": n_ep, "seconds_per_epoch": dt / n_ep, "value": n_ep * len(ds) / dt,
                        "unit": "scene-windows/s", "first_epoch_loss": float(ep_losses[0]),
                        "last_epoch_loss": float(last)}

    # ---- stand-alone HBM kernels on a working set beyond the Infinity Cache (rank 0) --------------------------------
    if rank == 0 and not args.no_extras and not args.kernels_only:
        vb = 32
        nb = 16384                                            # adjacency: 16384 x 8 x 32 x 32 x 4 B = 537 MB > 256 MiB
        rel_b, _ = synth_scenes(256, vb, 7)
        rel_b = torch.from_numpy(np.tile(rel_b, (nb // 256, 1, 1, 1))).to(dev)
        nodes_b, adj_b = ops.adj_build(rel_b)
        adj_ms = time_kernel(torch, lambda: ops.adj_build(rel_b, out=(nodes_b, adj_b)))
        agg_in = torch.randn(nb, 5, T_OBS, vb, device=dev)
        agg_ms = time_kernel(torch, lambda: ops.spatial_agg(agg_in, adj_b))
        agg_in.requires_grad_(True)
        yb = ops.spatial_agg(agg_in, adj_b)
        gy = torch.randn_like(yb)
        aggb_ms = time_kernel(torch, lambda: torch.autograd.grad(yb, agg_in, gy, retain_graph=True))
        del yb, gy
        adj_bytes = nb * (64 * vb + 32 * vb * vb + 64 * vb)
        agg_bytes = nb * (32 * vb * vb + 2 * 160 * vb)

        def row(ms, nbytes):
            return {"ms": ms, "GBps": nbytes / ms / 1e6, "frac_hbm": nbytes / ms / 1e6 / PEAK_HBM_GBS}
        out["kernels"] = {"working_set": "N=%d scene-windows, V=%d: adjacency %.0f MB (> 256 MiB Infinity Cache)"
                                         % (nb, vb, nb * 32 * vb * vb / 1e6),
                          "adj_build": row(adj_ms, adj_bytes), "spatial_agg_fwd": row(agg_ms, agg_bytes),
                          "spatial_agg_bwd": row(aggb_ms, agg_bytes)}
        del rel_b, nodes_b, adj_b, agg_in
        torch.cuda.empty_cache()

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            counts0 = sets[0][2]
            one, many = cpu_baselines(dsets[0]["v"] if counts0 is None else int(np.mean(counts0) + 0.5), args.cpu_seconds)
            out["cpu_baseline"] = one
            out["cpu_baseline_all_cores"] = many
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
