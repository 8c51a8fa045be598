"""CPU oracle for the Social-STGCNN hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement of the reference algorithm (reference repo
GRatTWCU/Social-STGCNN, mounted read-only at /root/reference while the build
runs).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it; the product (social_stgcnn_amd/) never does and has no CPU fallback.

Pinning: every function here is checked against fixtures in tests/golden/
that were produced by importing the reference itself (tests/golden/make_golden.py,
run once in the build container; the reference never travels to the GPU box).

The arithmetic is fp32 on torch CPU ops (conv2d / batch_norm / prelu /
matmul) because that is what the reference computes with; the adjacency build
is fp64 numpy cast to fp32, as in the reference.  One scene-window at a time
(N = 1), which is how the reference's train.py / test.py drive the model.

Reference map (file:line in /root/reference):
  anorm, seq_to_graph            utils.py:23-53 (+ networkx normalized_laplacian_matrix)
  ConvTemporalGraphical.forward  model.py:64-68
  st_gcn.forward                 model.py:145-155  (module layout model.py:92-143)
  social_stgcnn.forward          model.py:182-198  (module layout model.py:158-178)
  bivariate_loss                 metrics.py:84-113
  train() group semantics        train.py:28-79
  test() sampling / ADE / FDE    test.py:18-127, metrics.py:21-75
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------
# R1/R2: adjacency build
# --------------------------------------------------------------------------
def anorm(p1, p2):
    """utils.py:23-27 -- inverse Euclidean distance of two 2-vectors, 0 if equal."""
    d = math.sqrt((p1[0] - p2[0]) ** 2 + (p1[1] - p2[1]) ** 2)
    return 0 if d == 0 else 1 / d


def seq_to_graph_loops(seq_rel):
    """Pure-Python-loop restatement of utils.py:29-53 for small cases.

    seq_rel: torch fp32 tensor (V, 2, T) of relative displacements (the reference
    does the distance arithmetic on 0-dim fp32 torch tensors, utils.py:45).
    Returns fp32 torch tensors nodes (T, V, 2), lap (T, V, V).

    The symmetric-normalised Laplacian is written in closed form instead of
    calling networkx (utils.py:49-50; networkx 2.3 pinned in requirements.txt:90):
        d_i = sum_j a_ij  (self-loop weight 1 counted once)
        L = D^-1/2 (D - A) D^-1/2
    """
    seq_rel = torch.as_tensor(seq_rel, dtype=torch.float32)
    n_ped, _, seq_len = seq_rel.shape
    nodes = np.zeros((seq_len, n_ped, 2))
    lap = np.zeros((seq_len, n_ped, n_ped))
    for s in range(seq_len):
        step = seq_rel[:, :, s]
        a = np.zeros((n_ped, n_ped))
        for h in range(n_ped):
            nodes[s, h, :] = step[h].numpy()
            a[h, h] = 1
            for k in range(h + 1, n_ped):
                w = anorm(step[h], step[k])
                a[h, k] = w
                a[k, h] = w
        deg = a.sum(axis=1)
        with np.errstate(divide="ignore"):
            dinv = 1.0 / np.sqrt(deg)
        dinv[np.isinf(dinv)] = 0
        lap[s] = dinv[:, None] * (np.diag(deg) - a) * dinv[None, :]
    return (torch.from_numpy(nodes).type(torch.float),
            torch.from_numpy(lap).type(torch.float))


def seq_to_graph_np(seq_rel):
    """Vectorised numpy restatement of utils.py:29-53 (same maths as above).

    seq_rel: (V, 2, T) fp32 array-like.  Returns fp32 numpy nodes (T,V,2), lap (T,V,V).
    Differences are taken in fp32 (as the reference's tensor arithmetic does),
    squares/sqrt/division in fp64 (python floats in the reference).
    """
    rel = np.asarray(seq_rel, dtype=np.float32)
    n_ped = rel.shape[0]
    p = np.transpose(rel, (2, 0, 1))                      # (T, V, 2) fp32
    diff = p[:, :, None, :] - p[:, None, :, :]            # fp32 subtraction
    sq = (diff ** 2).astype(np.float32)                   # tensor**2 stays fp32
    ssum = (sq[..., 0] + sq[..., 1]).astype(np.float32)   # fp32 add
    dist = np.sqrt(ssum.astype(np.float64))               # math.sqrt on python float
    with np.errstate(divide="ignore"):
        a = np.where(dist == 0, 0.0, 1.0 / dist)
    eye = np.eye(n_ped, dtype=bool)[None]
    a = np.where(eye, 1.0, a)
    deg = a.sum(axis=2)
    dinv = 1.0 / np.sqrt(deg)
    lap = -a * dinv[:, :, None] * dinv[:, None, :]
    idx = np.arange(n_ped)
    lap[:, idx, idx] = (deg - 1.0) * dinv * dinv
    return p.astype(np.float32).copy(), lap.astype(np.float32)


# --------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------
def state_dict_keys(n_stgcnn=1, n_txpcnn=5):
    """The reference's state_dict key order (model.py:158-178)."""
    keys = []
    for j in range(n_stgcnn):
        s = "st_gcns.%d." % j
        keys += [s + "gcn.conv.weight", s + "gcn.conv.bias"]
        for bn in ("tcn.0",):
            keys += [s + bn + k for k in (".weight", ".bias", ".running_mean", ".running_var",
                                          ".num_batches_tracked")]
        keys += [s + "tcn.1.weight", s + "tcn.2.weight", s + "tcn.2.bias"]
        keys += [s + "tcn.3" + k for k in (".weight", ".bias", ".running_mean", ".running_var",
                                           ".num_batches_tracked")]
        keys += [s + "residual.0.weight", s + "residual.0.bias"]
        keys += [s + "residual.1" + k for k in (".weight", ".bias", ".running_mean",
                                                ".running_var", ".num_batches_tracked")]
        keys += [s + "prelu.weight"]
    for j in range(n_txpcnn):
        keys += ["tpcnns.%d.weight" % j, "tpcnns.%d.bias" % j]
    keys += ["tpcnn_ouput.weight", "tpcnn_ouput.bias"]
    keys += ["prelus.%d.weight" % j for j in range(n_txpcnn)]
    return keys


def clone_state(state):
    return {k: (v.clone() if torch.is_tensor(v) else torch.as_tensor(v).clone())
            for k, v in state.items()}


def _bn(x, state, prefix, training):
    """nn.BatchNorm2d (eps 1e-5, momentum .1, affine, running stats), model.py:114,123,140."""
    if training:
        state[prefix + ".num_batches_tracked"] += 1
    return F.batch_norm(x, state[prefix + ".running_mean"], state[prefix + ".running_var"],
                        state[prefix + ".weight"], state[prefix + ".bias"],
                        training, BN_MOMENTUM, BN_EPS)


# --------------------------------------------------------------------------
# R3/R4/R5: the model, one scene (or a same-V batch) at a time
# --------------------------------------------------------------------------
def conv_temporal_graphical(state, prefix, x, A):
    """model.py:64-68: 1x1 conv then out[n,c,t,w] = sum_v x[n,c,t,v] A[t,v,w].

    A may be (T,V,V) (reference) or (N,T,V,V) (north-star 'nctv,ntvw->nctw').
    """
    x = F.conv2d(x, state[prefix + ".conv.weight"], state[prefix + ".conv.bias"])
    if A.dim() == 3:
        y = torch.matmul(x.permute(0, 2, 1, 3), A.unsqueeze(0))       # (N,T,C,W)
    else:
        y = torch.matmul(x.permute(0, 2, 1, 3), A)
    return y.permute(0, 2, 1, 3).contiguous()


def st_gcn_forward(state, prefix, x, A, training, kt=3, use_mdn=False, residual="conv"):
    """model.py:145-155.  residual: 'conv' (Conv1x1+BN), 'identity' or 'zero' (model.py:127-141)."""
    if residual == "conv":
        res = F.conv2d(x, state[prefix + ".residual.0.weight"], state[prefix + ".residual.0.bias"])
        res = _bn(res, state, prefix + ".residual.1", training)
    elif residual == "identity":
        res = x
    else:
        res = 0
    g = conv_temporal_graphical(state, prefix + ".gcn", x, A)
    h = _bn(g, state, prefix + ".tcn.0", training)
    h = F.prelu(h, state[prefix + ".tcn.1.weight"])
    h = F.conv2d(h, state[prefix + ".tcn.2.weight"], state[prefix + ".tcn.2.bias"],
                 padding=((kt - 1) // 2, 0))
    h = _bn(h, state, prefix + ".tcn.3", training)
    out = h + res
    if not use_mdn:
        out = F.prelu(out, state[prefix + ".prelu.weight"])
    return out


def social_stgcnn_forward(state, x, A, training, n_stgcnn=1, n_txpcnn=5, kt=3):
    """model.py:182-198.  x (N,Cin,T,V); A (T,V,V) or (N,T,V,V).  Returns (N,C,P,V).

    BN running buffers in `state` are updated in place when training (one
    momentum update per call, like the reference's per-scene forward).
    The two .view() calls are memory reinterpretations, not transposes.
    """
    v = x
    for k in range(n_stgcnn):
        res_kind = "conv" if v.shape[1] != state["st_gcns.%d.gcn.conv.weight" % k].shape[0] \
            else "identity"
        v = st_gcn_forward(state, "st_gcns.%d" % k, v, A, training, kt=kt, residual=res_kind)
    v = v.contiguous()
    n, c, t, w = v.shape
    v = v.view(n, t, c, w)
    v = F.prelu(F.conv2d(v, state["tpcnns.0.weight"], state["tpcnns.0.bias"], padding=1),
                state["prelus.0.weight"])
    for k in range(1, n_txpcnn - 1):
        v = F.prelu(F.conv2d(v, state["tpcnns.%d.weight" % k], state["tpcnns.%d.bias" % k],
                             padding=1), state["prelus.%d.weight" % k]) + v
    v = F.conv2d(v, state["tpcnn_ouput.weight"], state["tpcnn_ouput.bias"], padding=1)
    n, p, c, w = v.shape
    return v.contiguous().view(n, c, p, w)


# --------------------------------------------------------------------------
# R6: loss
# --------------------------------------------------------------------------
def bivariate_loss(V_pred, V_trgt):
    """metrics.py:84-113: bivariate-Gaussian NLL, mean over (P, V).

    V_pred (P,V,5) = mux, muy, log sx, log sy, atanh rho ; V_trgt (P,V,2).
    """
    dx = V_trgt[..., 0] - V_pred[..., 0]
    dy = V_trgt[..., 1] - V_pred[..., 1]
    sx = torch.exp(V_pred[..., 2])
    sy = torch.exp(V_pred[..., 3])
    rho = torch.tanh(V_pred[..., 4])
    sxsy = sx * sy
    z = (dx / sx) ** 2 + (dy / sy) ** 2 - 2 * ((rho * dx * dy) / sxsy)
    one_m = 1 - rho ** 2
    pdf = torch.exp(-z / (2 * one_m)) / (2 * np.pi * (sxsy * torch.sqrt(one_m)))
    return torch.mean(-torch.log(torch.clamp(pdf, min=1e-20)))


def scene_loss(state, x, A, target, training, **cfg):
    """One reference training-loop body (train.py:48-59): forward + per-scene loss.

    x (1,2,T,V), A (T,V,V), target (P,V,2).  Returns (loss, V_pred (P,V,5))."""
    out = social_stgcnn_forward(state, x, A, training, **cfg)      # (1,5,P,V)
    v_pred = out.permute(0, 2, 3, 1)[0]
    return bivariate_loss(v_pred, target), v_pred


# --------------------------------------------------------------------------
# R9: one optimizer group of the reference loop
# --------------------------------------------------------------------------
def group_boundaries(n_scenes, batch_size):
    """train.py:34,58: 0-based indices of scenes that close a group (and are
    forwarded but NOT added to the loss)."""
    turn_point = int(n_scenes / batch_size) * batch_size + n_scenes % batch_size - 1
    return [i for i in range(n_scenes) if (i + 1) % batch_size == 0 or i == turn_point]


def train_group(state, scenes, batch_size, lr, param_keys, **cfg):
    """train.py:36-77 for ONE group: `scenes` = list of (x, A, target); the last
    scene closes the group (forwarded, BN stats updated, no loss).  SGD(lr)
    without momentum (train.py:197).  Returns (reported_loss, grads dict)."""
    params = {k: state[k].detach().clone().requires_grad_(True) for k in param_keys}
    work = dict(state)
    work.update(params)
    loss = None
    for i, (x, A, tgt) in enumerate(scenes):
        l, _ = scene_loss(work, x, A, tgt, True, **cfg)
        if i != len(scenes) - 1:
            loss = l if loss is None else loss + l
    loss = loss / batch_size
    loss.backward()
    grads = {}
    with torch.no_grad():
        for k in param_keys:
            g = params[k].grad
            grads[k] = None if g is None else g.clone()
            if g is not None:
                state[k] = (params[k] - lr * g).detach()
            else:
                state[k] = params[k].detach()
    return float(loss.item()), grads


# --------------------------------------------------------------------------
# R10: evaluation bookkeeping
# --------------------------------------------------------------------------
def rel_to_abs(rel, start):
    """metrics.py:66-75: cumulative sum of displacements + start position.
    rel (T,V,2), start (V,2) -> (T,V,2)."""
    return np.cumsum(rel, axis=0) + start[None]


def best_of_k_errors(v_pred, obs_abs_last, target_rel, k_steps, generator=None):
    """test.py:59-123 for one scene.  v_pred (P,V,5) torch fp32 (CPU); obs_abs_last (V,2);
    target_rel (P,V,2).  Samples k_steps trajectories from the predicted bivariate
    Gaussians (torch MultivariateNormal on the CPU default generator, as test.py does),
    returns per-ped (min ADE, min FDE) lists."""
    import torch.distributions.multivariate_normal as torchdist
    sx = torch.exp(v_pred[:, :, 2])
    sy = torch.exp(v_pred[:, :, 3])
    corr = torch.tanh(v_pred[:, :, 4])
    cov = torch.zeros(v_pred.shape[0], v_pred.shape[1], 2, 2)
    cov[:, :, 0, 0] = sx * sx
    cov[:, :, 0, 1] = corr * sx * sy
    cov[:, :, 1, 0] = corr * sx * sy
    cov[:, :, 1, 1] = sy * sy
    mvn = torchdist.MultivariateNormal(v_pred[:, :, 0:2], cov)
    tgt_abs = rel_to_abs(np.asarray(target_rel, dtype=np.float64), np.asarray(obs_abs_last))
    n_ped = v_pred.shape[1]
    ade = [[] for _ in range(n_ped)]
    fde = [[] for _ in range(n_ped)]
    for _ in range(k_steps):
        s = mvn.sample().numpy()
        s_abs = rel_to_abs(s.astype(np.float32), np.asarray(obs_abs_last, dtype=np.float32))
        err = np.sqrt(((s_abs.astype(np.float64) - tgt_abs) ** 2).sum(axis=2))   # (P,V)
        for n in range(n_ped):
            ade[n].append(float(err[:, n].mean()))
            fde[n].append(float(err[-1, n]))
    return [min(a) for a in ade], [min(f) for f in fde]


def best_of_k_errors_noise(v_pred, obs_abs_last, target_rel, eps):
    """`best_of_k_errors` with the standard-normal draws handed in: eps (K,P,V,2) torch fp32.
    MultivariateNormal.sample() is `loc + scale_tril @ standard_normal(shape)` with scale_tril =
    cholesky(cov) (test.py:59-71 builds cov, :89 samples), so with eps[k] = the k-th draw of the CPU
    generator this returns exactly what `best_of_k_errors` returns (pinned in tests/test_oracle.py)."""
    sx = torch.exp(v_pred[:, :, 2])
    sy = torch.exp(v_pred[:, :, 3])
    corr = torch.tanh(v_pred[:, :, 4])
    cov = torch.zeros(v_pred.shape[0], v_pred.shape[1], 2, 2)
    cov[:, :, 0, 0] = sx * sx
    cov[:, :, 0, 1] = corr * sx * sy
    cov[:, :, 1, 0] = corr * sx * sy
    cov[:, :, 1, 1] = sy * sy
    tril = torch.linalg.cholesky(cov)
    tgt_abs = rel_to_abs(np.asarray(target_rel, dtype=np.float64), np.asarray(obs_abs_last))
    n_ped = v_pred.shape[1]
    ade = [[] for _ in range(n_ped)]
    fde = [[] for _ in range(n_ped)]
    for k in range(eps.shape[0]):
        s = (v_pred[:, :, 0:2] + torch.matmul(tril, eps[k].unsqueeze(-1)).squeeze(-1)).numpy()
        s_abs = rel_to_abs(s.astype(np.float32), np.asarray(obs_abs_last, dtype=np.float32))
        err = np.sqrt(((s_abs.astype(np.float64) - tgt_abs) ** 2).sum(axis=2))
        for n in range(n_ped):
            ade[n].append(float(err[:, n].mean()))
            fde[n].append(float(err[-1, n]))
    return [min(a) for a in ade], [min(f) for f in fde]
