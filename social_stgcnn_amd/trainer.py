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

import numpy as np
import torch
import torch.distributed as dist

from . import ops
from .metrics import ade as _ade
from .metrics import bivariate_loss, fde as _fde, rel_to_abs


# ------------------------------------------------------------------------------------------
# distributed pieces (pure tensor functions: covered by gloo tests on CPU)
# ------------------------------------------------------------------------------------------
def allreduce_flat(flat, group=None):
    """Sum one flat gradient buffer over the ranks (a single collective call)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def fold_bn_across_ranks(before, after, n_local, momentum, group=None):
    """Exact sequential-fold of BatchNorm running statistics over ranks.

    Every rank updated its own copy `before -> after` with its n_local scenes:
        after = (1-m)^n_local * before + acc_r .
    The equivalent of ONE process seeing rank 0's scenes, then rank 1's, ... is
        r <- (1-m)^n_r * r + acc_r   for r = 0..R-1   (starting from the common `before`).
    Returns the folded statistics (same on every rank).
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return after
    world = dist.get_world_size(group)
    decay = (1.0 - momentum) ** int(n_local)
    acc = after - decay * before
    pack = torch.cat([acc.reshape(-1), torch.tensor([decay], dtype=acc.dtype, device=acc.device)])
    gathered = [torch.empty_like(pack) for _ in range(world)]
    dist.all_gather(gathered, pack, group=group)
    r = before.clone()
    for g in gathered:
        r = r * g[-1] + g[:-1].view_as(r)
    return r


def broadcast_module(model, src=0, group=None):
    """Initial parameter / buffer broadcast from rank `src`."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=src, group=group)


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

    def __init__(self, model, lr=0.01, clip_grad=None, group=None):
        self.model = model
        self.lr = lr
        self.clip_grad = clip_grad
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1

    def _bn_buffers(self):
        return self.model._tensors()[1]

    def forward_backward(self, x, adj, target, num_peds=None, weights=None):
        """One fused forward + loss + backward.  x (N,2,T,V) (any strides), adj (N,T,V,V) or (T,V,V),
        target (N,P,V,2).  Returns (weighted loss, per-scene losses, V_pred (N,5,P,V))."""
        model = self.model
        for p in model.parameters():
            p.grad = None
        y, _ = model(x, adj, num_peds)
        losses = bivariate_loss(y.permute(0, 2, 3, 1), target, num_peds)
        total = losses.sum() if weights is None else (losses * weights).sum()
        total.backward()
        return total.detach(), losses.detach(), y.detach()

    def step(self, x, adj, target, num_peds=None, weights=None):
        """forward_backward + gradient all-reduce + BatchNorm fold across ranks + SGD update."""
        model = self.model
        if self.world > 1:
            before = [b.detach().clone() for b in self._bn_buffers()]
        total, losses, y = self.forward_backward(x, adj, target, num_peds, weights)
        flat_p = model.flat_parameters()
        flat_g = self._flat_grad()
        if self.world > 1:
            allreduce_flat(flat_g, self.group)
            n_local = int(x.shape[0]) if num_peds is None else int((torch.as_tensor(num_peds) > 0).sum())
            mom = model.st_gcns[0].tcn[0].momentum
            for b, b0 in zip(self._bn_buffers(), before):
                b.copy_(fold_bn_across_ranks(b0, b.detach(), n_local, mom, self.group))
        if self.clip_grad is not None:
            torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], self.clip_grad)
            flat_g = self._flat_grad()
        ops.sgd_step(flat_p, flat_g, self.lr)
        return total, losses, y

    # ---- hipGraph capture of the step (launch-bound at these sizes: ~15 kernels of 10-150 us) ----------
    def capture(self, x, adj, target, num_peds=None, weights=None, warmup=3):
        """Capture forward + loss + backward (+ SGD update when single-rank) into ONE hipGraph on static
        input tensors and return `replay()`.  x / adj / target / num_peds / weights must be device tensors
        that stay alive; refresh their contents in place between replays.  With several ranks the graph
        holds forward+backward and the gradient all-reduce, BatchNorm fold and SGD run eagerly after it."""
        if num_peds is not None and not (torch.is_tensor(num_peds) and num_peds.is_cuda):
            raise ValueError("capture() needs num_peds as a device tensor (no host->device copies in a graph)")
        single = self.world == 1
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                if single:
                    self.step(x, adj, target, num_peds, weights)
                else:
                    self.forward_backward(x, adj, target, num_peds, weights)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self.step(x, adj, target, num_peds, weights) if single else \
                self.forward_backward(x, adj, target, num_peds, weights)
        model = self.model

        def replay():
            if single:
                graph.replay()
                return out
            before = model._pb.flat.clone()
            graph.replay()
            flat_g = self._flat_grad()
            allreduce_flat(flat_g, self.group)
            n_local = int(x.shape[0])
            mom = model.st_gcns[0].tcn[0].momentum
            model._pb.flat.copy_(fold_bn_across_ranks(before, model._pb.flat, n_local, mom, self.group))
            ops.sgd_step(model.flat_parameters(), flat_g, self.lr)
            return out
        self._graph = graph
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
            tgt_abs = rel_to_abs(tgt_rel[i, :, :c], obs_last[i, :c])
            a_ls = [[] for _ in range(c)]
            f_ls = [[] for _ in range(c)]
            for _ in range(k_steps):
                s_abs = rel_to_abs(mvn.sample().numpy(), obs_last[i, :c])
                for j in range(c):
                    a_ls[j].append(_ade(s_abs[:, j:j + 1], tgt_abs[:, j:j + 1]))
                    f_ls[j].append(_fde(s_abs[:, j:j + 1], tgt_abs[:, j:j + 1]))
            ades += [min(a) for a in a_ls]
            fdes += [min(f) for f in f_ls]
    return float(np.mean(ades)), float(np.mean(fdes)), ades, fdes
