"""Host-side ingest: ETH/UCY text files -> scene-windows (ragged) -> padded batches.

Counterpart of the reference's `TrajectoryDataset` windowing (utils.py:88-193):
`<frame> <ped> <x> <y>` rows, sliding `obs_len+pred_len`-frame windows with
`skip`, keep the pedestrians whose track spans the whole window, keep windows
with more than `min_ped` such pedestrians, absolute coordinates rounded to 4
decimals (utils.py:145) and relative = first difference with rel[0] = 0
(utils.py:153-155).  The per-window graphs are NOT built here: that is the
device kernel `adj_build` (social_stgcnn_amd.utils.seq_to_graph).

File names are sorted (the reference uses os.listdir order, which is
filesystem-dependent; window counts do not depend on the order).
"""
import math
import os

import numpy as np


def read_file(path, delim="\t"):
    """utils.py:72-83."""
    if delim == "tab":
        delim = "\t"
    elif delim == "space":
        delim = " "
    rows = []
    with open(path, "r") as f:
        for line in f:
            rows.append([float(tok) for tok in line.strip().split(delim)])
    return np.asarray(rows)


def poly_fit(traj, traj_len, threshold):
    """utils.py:56-71: 1.0 if a 2nd-order fit of the last traj_len points leaves a
    residual >= threshold (non-linear trajectory), else 0.0."""
    t = np.linspace(0, traj_len - 1, traj_len)
    rx = np.polyfit(t, traj[0, -traj_len:], 2, full=True)[1]
    ry = np.polyfit(t, traj[1, -traj_len:], 2, full=True)[1]
    return 1.0 if rx + ry >= threshold else 0.0


class SceneWindows:
    """Ragged set of scene-windows.

    seq, seq_rel : float64 (total_peds, 2, seq_len)   absolute / relative coordinates
    loss_mask    : float64 (total_peds, seq_len)
    non_linear   : float64 (total_peds,)
    seq_start_end: list of (start, end) into the ped axis, one per window (utils.py:189-193)
    """

    def __init__(self, seq, seq_rel, loss_mask, non_linear, num_peds, max_peds_in_frame):
        self.seq = seq
        self.seq_rel = seq_rel
        self.loss_mask = loss_mask
        self.non_linear = non_linear
        self.num_peds = np.asarray(num_peds, dtype=np.int64)
        cum = np.concatenate([[0], np.cumsum(self.num_peds)])
        self.seq_start_end = [(int(a), int(b)) for a, b in zip(cum[:-1], cum[1:])]
        self.max_peds_in_frame = max_peds_in_frame

    def __len__(self):
        return len(self.num_peds)


def load_windows(data_dir, obs_len=8, pred_len=12, skip=1, threshold=0.002, min_ped=1,
                 delim="\t", with_non_linear=True, files=None):
    """`files`: file names inside data_dir in the order to ingest them (default: sorted).  The reference
    walks os.listdir order (utils.py:116-117), which decides the order of the windows of a multi-file split."""
    seq_len = obs_len + pred_len
    names = sorted(os.listdir(data_dir)) if files is None else [str(f) for f in files]
    files = [os.path.join(data_dir, p) for p in names]
    seqs, rels, masks, nonlin, counts = [], [], [], [], []
    max_peds = 0
    for path in files:
        data = read_file(path, delim)
        frames = np.unique(data[:, 0])
        ped_ids = np.unique(data[:, 1])
        n_f, n_p = len(frames), len(ped_ids)
        f_idx = np.searchsorted(frames, data[:, 0])
        p_idx = np.searchsorted(ped_ids, data[:, 1])
        count = np.zeros((n_p, n_f), dtype=np.int32)
        np.add.at(count, (p_idx, f_idx), 1)
        pos = np.zeros((n_p, n_f, 2))
        pos[p_idx, f_idx] = np.around(data[:, 2:4], decimals=4)
        present = count > 0
        csum = np.concatenate([np.zeros((n_p, 1), dtype=np.int64), np.cumsum(count, axis=1)],
                              axis=1)
        num_sequences = int(math.ceil((n_f - seq_len + 1) / skip))
        for idx in range(0, num_sequences * skip + 1, skip):
            hi = min(idx + seq_len, n_f)
            if hi <= idx:
                continue
            rows_in_win = csum[:, hi] - csum[:, idx]
            in_win = rows_in_win > 0
            max_peds = max(max_peds, int(in_win.sum()))
            if hi - idx < seq_len:
                continue                     # no track can span the window (utils.py:148)
            # first / last frame of each ped inside the window (utils.py:146-147)
            win = present[:, idx:hi]
            first = np.argmax(win, axis=1)
            last = seq_len - 1 - np.argmax(win[:, ::-1], axis=1)
            keep = in_win & (first == 0) & (last == seq_len - 1)
            sel = np.nonzero(keep)[0]
            if np.any(rows_in_win[sel] != seq_len):
                # the reference would fail its slice assignment here (utils.py:156)
                raise ValueError("track with gaps or duplicate rows in %s window %d" % (path, idx))
            if len(sel) <= min_ped:
                continue
            cur = np.transpose(pos[sel, idx:hi, :], (0, 2, 1))          # (V, 2, seq_len)
            rel = np.zeros_like(cur)
            rel[:, :, 1:] = cur[:, :, 1:] - cur[:, :, :-1]
            seqs.append(cur)
            rels.append(rel)
            masks.append(np.ones((len(sel), seq_len)))
            if with_non_linear:
                nonlin += [poly_fit(c, pred_len, threshold) for c in cur]
            else:
                nonlin += [0.0] * len(sel)
            counts.append(len(sel))
    return SceneWindows(np.concatenate(seqs, axis=0), np.concatenate(rels, axis=0),
                        np.concatenate(masks, axis=0), np.asarray(nonlin), counts, max_peds)


def concat_windows(parts):
    """One SceneWindows holding the windows of `parts` in order (a split is the concatenation of its files' windows,
    utils.py:116-193; BASELINE configs[2] concatenates the five splits)."""
    return SceneWindows(np.concatenate([p.seq for p in parts], axis=0), np.concatenate([p.seq_rel for p in parts], axis=0),
                        np.concatenate([p.loss_mask for p in parts], axis=0),
                        np.concatenate([p.non_linear for p in parts], axis=0),
                        np.concatenate([p.num_peds for p in parts]), max(p.max_peds_in_frame for p in parts))


# The ETH/UCY leave-one-out protocol of the reference's datasets/<split>/train directories: eight recordings, every
# split trains on the ones that are not its test scene (datasets/*/train, listed with `ls`).
TRAIN_RECORDINGS = ("biwi_eth_train.txt", "biwi_hotel_train.txt", "crowds_zara01_train.txt", "crowds_zara02_train.txt",
                    "crowds_zara03_train.txt", "students001_train.txt", "students003_train.txt", "uni_examples_train.txt")
TRAIN_LEFT_OUT = {"eth": ("biwi_eth_train.txt",), "hotel": ("biwi_hotel_train.txt",),
                  "univ": ("students001_train.txt", "students003_train.txt"), "zara1": ("crowds_zara01_train.txt",),
                  "zara2": ("crowds_zara02_train.txt",)}


def load_train_splits(dirs, splits=("eth", "hotel", "univ", "zara1", "zara2"), obs_len=8, pred_len=12, skip=1,
                      with_non_linear=False):
    """The train sets of `splits` built from the eight recordings found in the directories `dirs` (every recording is
    ingested once; a window does not depend on which split it is in).  Returns {split: SceneWindows} with each split's
    recordings in sorted order."""
    where = {}
    for d in dirs:
        for name in sorted(os.listdir(d)):
            if name in TRAIN_RECORDINGS and name not in where:
                where[name] = d
    missing = [r for r in TRAIN_RECORDINGS if r not in where]
    if missing:
        raise FileNotFoundError("train recordings not found in %s: %s" % (list(dirs), missing))
    per_file = {r: load_windows(where[r], obs_len, pred_len, skip, with_non_linear=with_non_linear, files=[r])
                for r in TRAIN_RECORDINGS}
    return {s: concat_windows([per_file[r] for r in TRAIN_RECORDINGS if r not in TRAIN_LEFT_OUT[s]]) for s in splits}


def pad_batch(windows, indices, obs_len=8, v_pad=None):
    """Collate scene-windows `indices` into padded fp32 arrays (the layout the
    device kernels consume; padded pedestrian slots are zero and masked by num_peds):

    obs_rel  (N, obs_len, Vp, 2)   node features of the observed window  (V_obs, utils.py:41)
    pred_rel (N, pred_len, Vp, 2)  targets                               (V_tr)
    obs_abs  (N, obs_len, Vp, 2), pred_abs (N, pred_len, Vp, 2)
    num_peds (N,) int32
    """
    n = len(indices)
    counts = windows.num_peds[indices]
    vp = int(counts.max()) if v_pad is None else int(v_pad)
    seq_len = windows.seq.shape[2]
    pred_len = seq_len - obs_len
    obs_rel = np.zeros((n, obs_len, vp, 2), dtype=np.float32)
    pred_rel = np.zeros((n, pred_len, vp, 2), dtype=np.float32)
    obs_abs = np.zeros((n, obs_len, vp, 2), dtype=np.float32)
    pred_abs = np.zeros((n, pred_len, vp, 2), dtype=np.float32)
    for j, i in enumerate(indices):
        s, e = windows.seq_start_end[i]
        v = e - s
        rel = np.transpose(windows.seq_rel[s:e], (2, 0, 1))            # (seq_len, V, 2)
        ab = np.transpose(windows.seq[s:e], (2, 0, 1))
        obs_rel[j, :, :v] = rel[:obs_len]
        pred_rel[j, :, :v] = rel[obs_len:]
        obs_abs[j, :, :v] = ab[:obs_len]
        pred_abs[j, :, :v] = ab[obs_len:]
    return obs_rel, pred_rel, obs_abs, pred_abs, counts.astype(np.int32)
