"""Batched training / evaluation harness: the build's counterpart of the reference loops
train.train / train.vald (train.py:28-122) and test.test (test.py:18-127).

The reference pushes one scene at a time through the model and emulates a batch by summing
`batch_size - 1` per-scene losses before one backward (train.py:58-74).  Here a whole group is ONE
fused forward + loss + backward on the GPU; the semantics are kept exactly:
  * every scene of the group is forwarded (so BatchNorm running statistics see it),
  * the scene that closes the group is EXCLUDED from the loss (train.py:58),
  * the summed per-scene means are divided by `batch_size` whatever the group size (train.py:67),
  * SGD(lr) without momentum (train.py:197), optional clip_grad_norm_ (train.py:71-72).

Data parallelism: scene-windows are sharded over ranks, one flat 7,563-float gradient
all-reduce (RCCL over xGMI; a latency-bound 30 KB message) per optimizer step, and the BatchNorm
running statistics are folded across ranks in rank order so that R ranks x B scenes equal one rank
processing the concatenated R*B scenes.
"""
import math
import os

import numpy as np
import torch
import torch.distributed as dist

from . import ops
from .metrics import bivariate_loss, rel_to_abs


# ------------------------------------------------------------------------------------------
# distributed pieces (pure tensor functions: covered by gloo tests on CPU)
# ------------------------------------------------------------------------------------------
def allreduce_flat(flat, group=None):
    """Sum one flat gradient buffer over the ranks (a single collective call)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def bn_fold_scale(counts, rank, momentum):
    """Weights of the cross-rank BatchNorm fold for rank `rank` given every rank's scene count:
    (decay of this rank's own shard, factor its contribution is scaled by, total decay)."""
    keep = 1.0 - momentum
    own = keep ** int(counts[rank])
    later = keep ** int(sum(counts[rank + 1:]))
    total = keep ** int(sum(counts))
    return own, later, total


def fold_bn_across_ranks(before, after, n_local, momentum, group=None, counts=None):
    """Exact sequential-fold of BatchNorm running statistics over ranks with ONE all-reduce.

    Every rank updated its own copy `before -> after` with its n_local scenes:
        after = (1-m)^n_local * before + acc_r .
    ONE process seeing rank 0's scenes, then rank 1's, ... would end at
        before * prod_j (1-m)^n_j + sum_r acc_r * prod_{j>r} (1-m)^n_j ,
    so each rank scales its acc_r by the decay of the ranks after it and the sum is a plain all-reduce
    (the trainer packs it behind the gradient: one collective per step).  `counts` = scenes per rank
    (gathered when not given).  Returns the folded statistics (same on every rank)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return after
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if counts is None:
        t = torch.tensor([int(n_local)], dtype=torch.int64, device=after.device)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
        counts = [int(p.item()) for p in parts]
    own, later, total = bn_fold_scale(counts, rank, momentum)
    contrib = (after - own * before) * later
    dist.all_reduce(contrib, op=dist.ReduceOp.SUM, group=group)
    return before * total + contrib


def pack_rank_slot(flat_grad, bn_before, bn_after, n_nonempty, momentum, rank, world):
    """Pure-tensor statement of stg_dp_pack (the trainer launches the kernel; tests run this one under gloo on CPU):
    [gradient | world slots of (acc_r, n_r)], only this rank's slot filled."""
    n_p, n_b = flat_grad.numel(), bn_before.numel()
    pack = torch.zeros(n_p + world * (n_b + 1), dtype=torch.float32, device=flat_grad.device)
    pack[:n_p] = flat_grad
    own = (1.0 - momentum) ** int(n_nonempty)
    lo = n_p + rank * (n_b + 1)
    pack[lo:lo + n_b] = bn_after - own * bn_before
    pack[lo + n_b] = float(n_nonempty)
    return pack


def fold_from_pack(pack, bn_before, momentum, world, n_params):
    """Pure-tensor statement of stg_dp_fold: the running statistics one process would hold after seeing rank 0's
    scenes, then rank 1's, ... from the all-reduced pack."""
    n_b = bn_before.numel()
    keep = 1.0 - momentum
    slots = pack[n_params:].reshape(world, n_b + 1).double()
    counts = slots[:, n_b]
    later = torch.flip(torch.cumsum(torch.flip(counts, [0]), 0), [0]) - counts        # scenes of the ranks behind r
    acc = (slots[:, :n_b] * (keep ** later)[:, None]).sum(0)
    return (bn_before.double() * keep ** counts.sum() + acc).float()


def broadcast_module(model, src=0, group=None):
    """Initial parameter / buffer broadcast from rank `src`."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=src, group=group)


def shard_scenes(num_peds, world, rank, balanced=True):
    """Indices of the scenes rank `rank` of `world` takes from a global batch (data parallel over scene-windows).

    balanced=False: contiguous shards.  balanced=True: per-scene work grows with the crowd size, so the scenes are
    sorted by pedestrian count (stable, descending) and dealt to the ranks boustrophedon -- every rank gets the same
    number of scenes (a requirement of the captured multi-rank step) and near-equal summed crowd sizes.  The last
    len(num_peds) % world scenes are dropped (as a DistributedSampler with drop_last would)."""
    counts = np.asarray(num_peds).reshape(-1)
    per = len(counts) // world
    if not balanced:
        return np.arange(rank * per, (rank + 1) * per)
    order = np.argsort(-counts, kind="stable")[: per * world]
    rounds = order.reshape(per, world)
    rounds[1::2] = rounds[1::2, ::-1]                  # odd rounds run backwards over the ranks
    return np.sort(rounds[:, rank])


def group_weights(n_in_group, batch_size, device=None):
    """Per-scene loss weights of one reference group (train.py:58-67): 1/batch_size for every
    scene but the one that closes the group, which gets 0."""
    w = torch.full((n_in_group,), 1.0 / batch_size, dtype=torch.float32)
    w[-1] = 0.0
    return w if device is None else w.to(device)


def group_bounds(n_scenes, batch_size):
    """0-based index of the scene closing each group (train.py:34,58)."""
    turn_point = int(n_scenes / batch_size) * batch_size + n_scenes % batch_size - 1
    return [i for i in range(n_scenes) if (i + 1) % batch_size == 0 or i == turn_point]


# ------------------------------------------------------------------------------------------
class Trainer:
    """SGD trainer over the fused HIP path."""

    def __init__(self, model, lr=0.01, clip_grad=None, group=None, lr_sh_rate=None, lr_gamma=0.2):
        """lr / clip_grad as train.py:152-156,197; lr_sh_rate = StepLR step size in epochs (train.py:200,
        `--lr_sh_rate`, used when `--use_lrschd`), lr_gamma its decay (0.2 in the reference)."""
        self.model = model
        self.base_lr = float(lr)
        self.lr = float(lr)
        self.clip_grad = clip_grad
        self.group = group
        self.lr_sh_rate = lr_sh_rate
        self.lr_gamma = lr_gamma
        self.epoch = 0
        self._lr_dev = None                  # device copy of lr read by the update kernel (graph-safe)
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1

    def _bn_buffers(self):
        return self.model._tensors()[1]

    def _lr_tensor(self, device):
        if self._lr_dev is None or self._lr_dev.device != device:
            self._lr_dev = torch.full((1,), self.lr, device=device, dtype=torch.float32)
        return self._lr_dev

    def scheduler_step(self):
        """StepLR.step() (train.py:200,222-223): once per epoch; lr = base_lr * gamma ** (epoch // lr_sh_rate).
        The device copy is refreshed in place, so captured steps follow the schedule."""
        self.epoch += 1
        if self.lr_sh_rate:
            self.lr = self.base_lr * self.lr_gamma ** (self.epoch // int(self.lr_sh_rate))
            if self._lr_dev is not None:
                self._lr_dev.fill_(self.lr)
        return self.lr

    def _update(self, flat_p, flat_g):
        """clip_grad_norm_ + SGD update over the flat buffers: one launch (train.py:71-74)."""
        ops.optim_step(flat_p, flat_g, self.lr, self.clip_grad, self._lr_tensor(flat_p.device))

    def forward_backward(self, x, adj, target, num_peds=None, weights=None, defer_tail=False):
        """One fused forward + loss + backward.  x (N,2,T,V) (any strides), adj (N,T,V,V) or (T,V,V),
        target (N,P,V,2).  Returns (weighted loss, per-scene losses, V_pred (N,5,P,V)).  defer_tail: leave the loss
        total to the caller's tail launch (returned as None)."""
        model = self.model
        for p in model.parameters():
            p.grad = None
        y, _ = model(x, adj, num_peds)
        # loss and backward in the backward's own launches (the input stage of the wave-per-scene kernel computes
        # dV_pred from V_pred and the target: stg_model_bwd_nll); on the workgroup path: loss + its gradient w.r.t.
        # V_pred in ONE kernel (per-scene weights folded in), then backward straight from dV_pred -- either way no
        # autograd graph through the loss, no separate scale / sum / expand kernels
        losses = ops.backward_from_target(model, y.detach(), target, weights) if y.requires_grad else None
        if losses is None:
            losses, dy = ops.bivariate_nll_with_grad(y.detach(), target, num_peds, weights)
            y.backward(dy)
        if defer_tail:
            return None, losses, y.detach()
        total = ops.weighted_sum(losses, weights)
        return total, losses, y.detach()

    def _dp_buffers(self, flat_p, flat_b):
        """the step's ONE collective buffer [gradient | world x (BatchNorm contribution, scene count)] + a snapshot
        slot for the running statistics (allocated once: the captured step replays on them)."""
        n = flat_p.numel() + self.world * (flat_b.numel() + 1)
        if getattr(self, "_pack", None) is None or self._pack.numel() != n or self._pack.device != flat_p.device:
            self._pack = torch.zeros(n, device=flat_p.device, dtype=torch.float32)
            self._bn_before = torch.empty_like(flat_b)
        return self._pack, self._bn_before

    def step(self, x, adj, target, num_peds=None, weights=None):
        """forward_backward + [ONE all-reduce: gradient and BatchNorm fold across ranks] + SGD update.  No host
        synchronisation anywhere: the scene counts the fold needs travel inside the collective."""
        model = self.model
        if self.world == 1 and self.clip_grad is None and model.training:
            # no clipping: the whole tail (SGD, running-statistics fold, reported loss) rides in the backward's last
            # launch (stg_model_bwd_step) -- when the batch runs the wave-per-scene kernels
            for p in model.parameters():
                p.grad = None
            model._defer_bn_fold, model._pending_bn = True, None
            try:
                y, _ = model(x, adj, num_peds)
            finally:
                model._defer_bn_fold = False
            flat_p = model.flat_parameters()
            res = None
            if y.requires_grad and model._pending_bn is not None:
                res = ops.backward_from_target(model, y.detach(), target, weights,
                                               step=(model._pending_bn, self.lr, self._lr_tensor(flat_p.device)))
            if res is not None:
                model._pending_bn = None
                return res[1], res[0], y.detach()
            # workgroup path: loss kernel + autograd backward + the tail launch
            losses, dy = ops.bivariate_nll_with_grad(y.detach(), target, num_peds, weights)
            y.backward(dy)
            total = ops.train_tail(model._pending_bn, losses, weights, flat_p, self._flat_grad(), self.lr, None,
                                   self._lr_tensor(flat_p.device))
            model._pending_bn = None
            return total, losses, y.detach()
        if self.world == 1:
            # forward + loss + backward, then ONE tail launch: running-statistics fold, reported loss, clip + SGD
            model._defer_bn_fold, model._pending_bn = model.training, None
            try:
                _, losses, y = self.forward_backward(x, adj, target, num_peds, weights, defer_tail=model.training)
            finally:
                model._defer_bn_fold = False
            flat_p = model.flat_parameters()
            if model._pending_bn is None:                       # (eval-mode model: nothing to fold)
                total = ops.weighted_sum(losses, weights)
                self._update(flat_p, self._flat_grad())
                return total, losses, y
            total = ops.train_tail(model._pending_bn, losses, weights, flat_p, self._flat_grad(), self.lr,
                                   self.clip_grad, self._lr_tensor(flat_p.device))
            model._pending_bn = None
            return total, losses, y
        flat_p = model.flat_parameters()
        flat_b = model._pb.ensure(self._bn_buffers())
        pack, before = self._dp_buffers(flat_p, flat_b)
        before.copy_(flat_b)
        total, losses, y = self.forward_backward(x, adj, target, num_peds, weights)
        mom = model.st_gcns[0].tcn[0].momentum
        rank = dist.get_rank(self.group)
        ops.dp_pack(self._flat_grad(), before, flat_b, num_peds, int(x.shape[0]), mom, rank, self.world, pack)
        dist.all_reduce(pack, op=dist.ReduceOp.SUM, group=self.group)          # the step's ONE collective
        ops.dp_fold(pack, before, mom, rank, self.world, flat_p.numel(), flat_b, model._tensors()[2])
        self._update(flat_p, pack[:flat_p.numel()])
        return total, losses, y

    # ---- hipGraph capture of the step (launch-bound at these sizes: ~15 kernels of 10-150 us) ----------
    def capture(self, x, adj, target, num_peds=None, weights=None, warmup=3, pre=None):
        """Capture the training step on static input tensors and return `replay()`.

        Single rank: ONE hipGraph = forward + loss + backward + SGD update.
        Several ranks: graph A = snapshot of the BatchNorm statistics + forward + loss + backward + stg_dp_pack
        ([gradient | this rank's BatchNorm contribution and scene count] in one flat buffer); then ONE eager
        all-reduce (RCCL) of that buffer (7,563 + world x 31 floats); graph B = stg_dp_fold + SGD update.
        x / adj / target / num_peds / weights must be device tensors that stay alive; refresh their contents in place
        between replays.  `pre` (optional callable) runs inside the graph ahead of the step -- e.g. the adjacency
        build that fills x / adj from relative trajectories."""
        if num_peds is not None and not (torch.is_tensor(num_peds) and num_peds.is_cuda):
            raise ValueError("capture() needs num_peds as a device tensor (no host->device copies in a graph)")
        single = self.world == 1
        model = self.model
        # warm-up launches (outside the capture) must not change the model: snapshot / restore its state
        flat_p = model.flat_parameters()
        flat_b = model._pb.ensure(self._bn_buffers())
        nbt = model._tensors()[2]
        self._lr_tensor(flat_p.device)       # allocate outside the capture
        saved = (flat_p.clone(), flat_b.clone(), [t.clone() for t in nbt])
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                if pre is not None:
                    pre()
                if single:
                    self.step(x, adj, target, num_peds, weights)
                else:
                    self.forward_backward(x, adj, target, num_peds, weights)
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad():
            flat_p.copy_(saved[0])
            flat_b.copy_(saved[1])
            for t, t0 in zip(nbt, saved[2]):
                t.copy_(t0)
        if single:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                if pre is not None:
                    pre()
                out = self.step(x, adj, target, num_peds, weights)
            self._graph = graph

            def replay():
                graph.replay()
                return out
            return replay

        n_p = flat_p.numel()
        rank = dist.get_rank(self.group)
        mom = model.st_gcns[0].tcn[0].momentum
        pack, before = self._dp_buffers(flat_p, flat_b)
        g_a, g_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_a):
            if pre is not None:
                pre()
            before.copy_(flat_b)
            out = self.forward_backward(x, adj, target, num_peds, weights)
            ops.dp_pack(self._flat_grad(), before, flat_b, num_peds, int(x.shape[0]), mom, rank, self.world, pack)
        with torch.cuda.graph(g_b, pool=g_a.pool()):
            ops.dp_fold(pack, before, mom, rank, self.world, n_p, flat_b, nbt)
            self._update(flat_p, pack[:n_p])
        self._graph = (g_a, g_b)

        def replay():
            g_a.replay()
            dist.all_reduce(pack, op=dist.ReduceOp.SUM, group=self.group)      # the step's ONE collective
            g_b.replay()
            return out
        return replay

    def _flat_grad(self):
        """Gradients as one flat buffer in parameter order (dead parameters contribute zeros)."""
        flat = getattr(self.model, "_flat_grad", None)
        params = list(self.model._tensors()[0])
        if flat is not None:
            off, ok = 0, True
            for p in params:
                if p.grad is not None and p.grad.data_ptr() != flat.data_ptr() + 4 * off:
                    ok = False
                    break
                off += p.numel()
            if ok:
                return flat
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
        return flat

    # ---- reference-style epoch (train.py:28-79) -------------------------------------------------
    def train_epoch(self, batcher, n_scenes, batch_size):
        """`batcher(lo, hi)` -> (x, adj, target, num_peds) for scenes [lo, hi) of the epoch order.
        Returns the epoch loss exactly as train() reports it (sum of group losses / scenes seen)."""
        self.model.train()
        loss_sum, lo = 0.0, 0
        for b in group_bounds(n_scenes, batch_size):
            x, adj, target, peds = batcher(lo, b + 1)
            w = group_weights(b + 1 - lo, batch_size, x.device)
            total, _, _ = self.step(x, adj, target, peds, w)
            loss_sum += float(total)
            lo = b + 1
        return loss_sum / n_scenes

    @torch.no_grad()
    def val_epoch(self, batcher, n_scenes, batch_size):
        """vald() (train.py:81-122): model.eval(), same grouping, no update."""
        self.model.eval()
        loss_sum, lo = 0.0, 0
        for b in group_bounds(n_scenes, batch_size):
            x, adj, target, peds = batcher(lo, b + 1)
            y, _ = self.model(x, adj, peds)
            losses = bivariate_loss(y.permute(0, 2, 3, 1), target, peds)
            loss_sum += float((losses * group_weights(b + 1 - lo, batch_size, x.device)).sum())
            lo = b + 1
        return loss_sum / n_scenes


# ------------------------------------------------------------------------------------------
# evaluation (test.py:18-127)
# ------------------------------------------------------------------------------------------
@torch.no_grad()
def evaluate_ade_fde(model, batches, k_steps=20):
    """Best-of-k ADE/FDE like test.test(): the forward runs batched on the GPU, the sampling and the
    displacement bookkeeping follow the reference on the CPU (torch MultivariateNormal on the default
    CPU generator, one scene at a time in order, so a seed reproduces the reference's draws).

    batches: iterable of (x, adj, num_peds, obs_abs_last (N,V,2), target_rel (N,P,V,2)) device/CPU tensors.
    Returns (ade, fde, per_ped_ade, per_ped_fde)."""
    import torch.distributions.multivariate_normal as torchdist
    model.eval()
    ades, fdes = [], []
    for x, adj, peds, obs_last, tgt_rel in batches:
        y, _ = model(x, adj, peds)
        v_pred = y.permute(0, 2, 3, 1).cpu()                      # (N,P,V,5)
        n = v_pred.shape[0]
        counts = [v_pred.shape[2]] * n if peds is None else [int(c) for c in torch.as_tensor(peds).cpu()]
        obs_last = torch.as_tensor(obs_last).cpu().numpy()
        tgt_rel = torch.as_tensor(tgt_rel).cpu().numpy()
        for i in range(n):
            c = counts[i]
            vp = v_pred[i, :, :c]
            sx, sy, corr = torch.exp(vp[:, :, 2]), torch.exp(vp[:, :, 3]), torch.tanh(vp[:, :, 4])
            cov = torch.zeros(vp.shape[0], c, 2, 2)
            cov[:, :, 0, 0] = sx * sx
            cov[:, :, 0, 1] = corr * sx * sy
            cov[:, :, 1, 0] = corr * sx * sy
            cov[:, :, 1, 1] = sy * sy
            mvn = torchdist.MultivariateNormal(vp[:, :, 0:2], cov)
            tgt_abs = rel_to_abs(tgt_rel[i, :, :c], obs_last[i, :c]).astype(np.float64)
            # k sequential draws (the reference draws inside its sample loop, test.py:87-89), then the displacement
            # bookkeeping of metrics.py:21-53 for all samples and pedestrians at once
            s_abs = np.stack([rel_to_abs(mvn.sample().numpy(), obs_last[i, :c]) for _ in range(k_steps)])
            err = np.sqrt(((s_abs.astype(np.float64) - tgt_abs[None]) ** 2).sum(axis=3))       # (k, P, c)
            ades += err.mean(axis=1).min(axis=0).tolist()
            fdes += err[:, -1].min(axis=0).tolist()
    return float(np.mean(ades)), float(np.mean(fdes)), ades, fdes


@torch.no_grad()
def evaluate_ade_fde_device(model, batches, k_steps=20, seed=0, noise_fn=None):
    """test.test() (test.py:18-127) with the sampling and the best-of-k displacement errors on the device
    (`ops.best_of_k`: one launch per batch instead of O(k V P) Python loops per scene).

    batches: as for `evaluate_ade_fde`.  The draws come from the kernel's Philox stream keyed by `seed` + batch
    index, or from `noise_fn(batch_index, (k, N, P, V, 2))` -> standard normals (parity tests feed the CPU
    sampler's numbers).  Statistically, not bitwise, equal to the reference's CPU draws.
    Returns (ade, fde, per_ped_ade, per_ped_fde)."""
    model.eval()
    ades, fdes = [], []
    for b, (x, adj, peds, obs_last, tgt_rel) in enumerate(batches):
        y, _ = model(x, adj, peds)
        n, _, p, v = y.shape
        dev = y.device
        tgt_rel = torch.as_tensor(tgt_rel).to(dev)
        obs_last = None if obs_last is None else torch.as_tensor(obs_last).to(dev)
        noise = noise_fn(b, (k_steps, n, p, v, 2)) if noise_fn is not None else None
        a, f = ops.best_of_k(y, tgt_rel, obs_last, peds, k_steps, noise, seed + b)
        if peds is None:
            mask = torch.ones((n, v), dtype=torch.bool, device=dev)
        else:
            counts = torch.as_tensor(peds).to(device=dev, dtype=torch.int64)
            mask = torch.arange(v, device=dev)[None, :] < counts[:, None]
        ades.append(a[mask])
        fdes.append(f[mask])
    ades = torch.cat(ades).cpu().numpy().astype(np.float64)
    fdes = torch.cat(fdes).cpu().numpy().astype(np.float64)
    return float(ades.mean()), float(fdes.mean()), ades.tolist(), fdes.tolist()


# ------------------------------------------------------------------------------------------
# checkpoint files of train.py:202-246 (what test.py:134-160 reads back)
# ------------------------------------------------------------------------------------------
class Checkpoint:
    """`<dir>/val_best.pth` (state_dict, 40 keys), `args.pkl`, `metrics.pkl`, `constant_metrics.pkl` with the
    bookkeeping of train.py:213-246: `record(epoch, train_loss, val_loss)` appends the losses, saves the model
    when the validation loss improves and rewrites the two metric files."""

    def __init__(self, directory, args=None):
        import pickle
        self.dir = directory
        os.makedirs(directory, exist_ok=True)
        self.metrics = {"train_loss": [], "val_loss": []}
        self.constant_metrics = {"min_val_epoch": -1, "min_val_loss": 9999999999999999}
        if args is not None:
            with open(os.path.join(directory, "args.pkl"), "wb") as fp:
                pickle.dump(args, fp)

    def record(self, epoch, model, train_loss, val_loss):
        import pickle
        self.metrics["train_loss"].append(train_loss)
        self.metrics["val_loss"].append(val_loss)
        improved = val_loss < self.constant_metrics["min_val_loss"]
        if improved:
            self.constant_metrics["min_val_loss"] = val_loss
            self.constant_metrics["min_val_epoch"] = epoch
            state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
            torch.save(state, os.path.join(self.dir, "val_best.pth"))
        with open(os.path.join(self.dir, "metrics.pkl"), "wb") as fp:
            pickle.dump(self.metrics, fp)
        with open(os.path.join(self.dir, "constant_metrics.pkl"), "wb") as fp:
            pickle.dump(self.constant_metrics, fp)
        return improved


def load_checkpoint(model, path, map_location="cpu"):
    """model.load_state_dict(torch.load(val_best.pth)) (test.py:158) with the loader that executes nothing from
    the file."""
    state = torch.load(path, map_location=map_location, weights_only=True)
    model.load_state_dict(state)
    return model
