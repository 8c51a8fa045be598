"""Drop-in `metrics.py` hot-path function: bivariate_loss on the fused `nll` HIP kernel
(reference metrics.py:84-113), plus the evaluation bookkeeping of metrics.py:21-75 in
vectorised host form for the repo's own eval harness."""
import numpy as np
import torch

from . import ops


def bivariate_loss(V_pred, V_trgt, num_peds=None):
    """metrics.py:84-113.  Reference call: V_pred (P,V,5), V_trgt (P,V,2) -> scalar (mean over P,V).
    Batched call: V_pred (N,P,V,5), V_trgt (N,P,V,2) [+ num_peds] -> per-scene losses (N,).
    V_pred may be any strided view (e.g. the model output permuted like train.py:52)."""
    if V_pred.dim() == 3:
        return ops.bivariate_nll(V_pred.unsqueeze(0), V_trgt.unsqueeze(0), None)[0]
    return ops.bivariate_nll(V_pred, V_trgt, num_peds)


def rel_to_abs(nodes, init_node):
    """metrics.py:66-75 (nodes_rel_to_nodes_abs): cumulative displacements + start position."""
    nodes = np.asarray(nodes)
    return np.cumsum(nodes, axis=0, dtype=nodes.dtype) + np.asarray(init_node)[None]


def ade(pred, target):
    """metrics.py:21-37 for one scene: mean over peds and time of the displacement error. (T,V,2)."""
    err = np.sqrt(((np.asarray(pred, np.float64) - np.asarray(target, np.float64)) ** 2).sum(axis=2))
    return float(err.mean())


def fde(pred, target):
    """metrics.py:40-53 for one scene: mean over peds of the final-step error."""
    err = np.sqrt(((np.asarray(pred, np.float64)[-1] - np.asarray(target, np.float64)[-1]) ** 2).sum(axis=1))
    return float(err.mean())
