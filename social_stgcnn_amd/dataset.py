"""Device-resident windowed dataset: the counterpart of the reference's TrajectoryDataset + DataLoader
(utils.py:88-193, train.py:167-177) with the windows living in HBM and batches gathered by index on the device.

    win = data.load_windows(dir)                 # host ingest (vectorised NumPy, bit-equal to the reference)
    ds = DeviceWindows(win, device)              # ONE upload: 160 bytes per pedestrian-window
    order = ds.shuffled_order(generator)         # device permutation of the windows (DataLoader(shuffle=True))
    batch = ds.gather(order[lo:hi])              # obs_rel / target / num_peds padded to ds.v_max, on the device

`EpochRunner` captures gather -> adj_build -> forward -> loss -> backward -> update as one hipGraph on static buffers
and replays it per group with the index buffer refreshed in place: a training epoch moves nothing over PCIe.
"""
import numpy as np
import torch

from . import ops
from ._lib import check, lib, ptr, require_gpu, stream_ptr


class DeviceWindows:
    def __init__(self, windows, device, obs_len=8):
        self.device = torch.device(device)
        self.n_windows = len(windows)
        self.t_all = int(windows.seq_rel.shape[2])
        self.t_obs = int(obs_len)
        self.t_pred = self.t_all - self.t_obs
        self.num_peds_host = np.asarray(windows.num_peds, dtype=np.int32)
        self.v_max = int(self.num_peds_host.max())
        start = np.concatenate([[0], np.cumsum(self.num_peds_host)]).astype(np.int32)
        self.rel_all = torch.from_numpy(np.ascontiguousarray(windows.seq_rel, dtype=np.float32)).to(self.device)
        self.win_start = torch.from_numpy(start).to(self.device)

    def __len__(self):
        return self.n_windows

    def shuffled_order(self, generator=None):
        """a device permutation of the window indices (int32)"""
        return torch.randperm(self.n_windows, device=self.device, generator=generator).to(torch.int32)

    def buffers(self, n, v_pad=None):
        v = int(v_pad or self.v_max)
        dev = self.device
        return (torch.empty((n, v, 2, self.t_obs), device=dev), torch.empty((n, self.t_pred, v, 2), device=dev),
                torch.empty(n, device=dev, dtype=torch.int32))

    def gather(self, index=None, n=None, v_pad=None, out=None):
        """windows `index` (int32 device tensor; None = the first n) -> (obs_rel (N,V,2,T_obs), target (N,T_pred,V,2),
        num_peds (N)) zero-padded to v_pad (default: the dataset's largest crowd).  One launch, no host sync."""
        if index is not None:
            require_gpu(index)
            if index.dtype != torch.int32 or not index.is_contiguous():
                raise TypeError("index must be a contiguous int32 device tensor")
            n = index.numel()
        obs_rel, target, peds = out if out is not None else self.buffers(n, v_pad)
        v = obs_rel.shape[1]
        check(lib().stg_gather_windows(ptr(self.rel_all), ptr(self.win_start), ptr(index), self.n_windows, int(n), int(v),
                                       self.t_obs, self.t_pred, ptr(obs_rel), ptr(target), ptr(peds), stream_ptr()),
              "stg_gather_windows")
        return obs_rel, target, peds


class EpochRunner:
    """Reference-style training epochs (train.py:28-79: groups of `batch_size` scenes, the closing scene forwarded but
    not in the loss) over a DeviceWindows dataset, every group ONE replay of a captured hipGraph:
    gather (by device index) -> adj_build -> forward -> NLL -> backward -> clip/SGD.  The short group at the end of an
    epoch has a graph of its own (captured on leading slices of the same static buffers), so an epoch is graph replays and
    device-to-device index copies only."""

    def __init__(self, trainer, dataset, batch_size, v_pad=None):
        self.trainer, self.ds, self.bs = trainer, dataset, int(batch_size)
        dev = dataset.device
        self.index = torch.zeros(self.bs, device=dev, dtype=torch.int32)
        # pedestrian slots: the dataset's largest crowd rounded up to a multiple of 4 -- rows of the adjacency are then
        # 16-byte aligned and stg_adj_build / the aggregation kernel move them with 16-byte accesses
        v_pad = int(v_pad) if v_pad else (dataset.v_max + 3) & ~3
        self.obs_rel, self.target, self.peds = dataset.buffers(self.bs, v_pad)
        v = self.obs_rel.shape[1]
        self.nodes = torch.empty((self.bs, dataset.t_obs, v, 2), device=dev)
        self.adj = torch.empty((self.bs, dataset.t_obs, v, v), device=dev)
        self._replays = {}                 # group size -> (replay, index slice, loss weights)

    def _group(self, cnt):
        """the captured step of a group of `cnt` scenes (cnt <= batch_size): leading slices of the static buffers"""
        g = self._replays.get(cnt)
        if g is None:
            from .trainer import group_weights
            index = self.index[:cnt]
            obs_rel, target, peds = self.obs_rel[:cnt], self.target[:cnt], self.peds[:cnt]
            nodes, adj = self.nodes[:cnt], self.adj[:cnt]
            weights = group_weights(cnt, self.bs, self.ds.device)

            def pre():
                self.ds.gather(index, out=(obs_rel, target, peds))
                ops.adj_build(obs_rel, peds, out=(nodes, adj))
            replay = self.trainer.capture(nodes.permute(0, 3, 1, 2), adj, target, peds, weights, pre=pre)
            # (the graph reads `weights` at every replay: it must stay alive with the graph)
            g = self._replays[cnt] = (replay, index, weights)
        return g

    def train_epoch(self, order):
        """order: int32 device tensor, a permutation (or any list) of window indices.  Returns the epoch loss as
        train() reports it (sum of group losses / scenes seen), accumulated on the device."""
        from .trainer import group_bounds
        self.trainer.model.train()
        n_scenes = order.numel()
        total = torch.zeros((), device=self.ds.device)
        lo = 0
        for b in group_bounds(n_scenes, self.bs):
            cnt = b + 1 - lo
            if cnt not in self._replays:
                self.index[:cnt].copy_(order[lo:b + 1])         # (valid window indices while the group is captured)
            replay, index, _ = self._group(cnt)
            index.copy_(order[lo:b + 1])                        # device -> device
            total = total + replay()[0]
            lo = b + 1
        return total / n_scenes
