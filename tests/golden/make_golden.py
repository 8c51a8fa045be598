#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE itself.

Run ONCE in the build container (the only place /root/reference exists):

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference (pure Python on torch/numpy/networkx) is imported read-only from
/root/reference; nothing from it is copied: the outputs are DATA (inputs and
expected outputs as fp32/fp64 arrays) that pin oracle/stgcnn_oracle.py and, through
it, the HIP path.  The shipped checkpoint is loaded with weights_only=True.
"""
import argparse
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)

import numpy as np
import torch

import model as ref_model          # /root/reference/model.py
import utils as ref_utils          # /root/reference/utils.py
import metrics as ref_metrics      # /root/reference/metrics.py

from social_stgcnn_amd import data as my_data

torch.set_num_threads(1)
CFG = dict(n_stgcnn=1, n_txpcnn=5, output_feat=5, seq_len=8, kernel_size=3, pred_seq_len=12)


def sd_to_np(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def new_ref_model(seed=None, weights=None):
    if seed is not None:
        torch.manual_seed(seed)
    m = ref_model.social_stgcnn(**CFG)
    if weights is not None:
        m.load_state_dict(weights)
    return m


def scene_from_windows(win, i):
    """-> obs_traj (V,2,8), pred_traj, obs_rel, pred_rel as fp32 torch (TrajectoryDataset.__getitem__)."""
    s, e = win.seq_start_end[i]
    seq = torch.from_numpy(win.seq[s:e]).type(torch.float)
    rel = torch.from_numpy(win.seq_rel[s:e]).type(torch.float)
    return seq[:, :, :8], seq[:, :, 8:], rel[:, :, :8], rel[:, :, 8:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=HERE)
    args = ap.parse_args()
    out = args.out

    # ---------------------------------------------------------------- weights
    eth_sd = torch.load(os.path.join(REF, "checkpoint/social-stgcnn-eth/val_best.pth"),
                        map_location="cpu", weights_only=True)
    np.savez(os.path.join(out, "weights_eth.npz"), **sd_to_np(eth_sd))

    # seed-0 default initialisation of the reference modules (pins constructor RNG order)
    m0 = new_ref_model(seed=0)
    np.savez(os.path.join(out, "init_seed0.npz"), **sd_to_np(m0.state_dict()))

    # ---------------------------------------------------------------- ingest
    # reference TrajectoryDataset on the two small test splits vs this repo's ingest
    ingest = {}
    for name in ("eth", "hotel"):
        d = os.path.join(REF, "datasets", name, "test") + "/"
        ds = ref_utils.TrajectoryDataset(d, obs_len=8, pred_len=12, skip=1, norm_lap_matr=True)
        mine = my_data.load_windows(d, 8, 12, 1)
        assert mine.seq_start_end == ds.seq_start_end, name
        full = torch.cat([ds.obs_traj, ds.pred_traj], dim=2).numpy()
        full_rel = torch.cat([ds.obs_traj_rel, ds.pred_traj_rel], dim=2).numpy()
        assert np.array_equal(mine.seq.astype(np.float32), full), name
        assert np.array_equal(mine.seq_rel.astype(np.float32), full_rel), name
        assert np.array_equal(mine.non_linear.astype(np.float32), ds.non_linear_ped.numpy()), name
        assert mine.max_peds_in_frame == ds.max_peds_in_frame
        ingest[name] = (ds, mine)
        print("ingest ok", name, len(ds), "windows", full.shape[0], "peds")
    ds_eth, win_eth = ingest["eth"]
    np.savez(os.path.join(out, "eth_test_windows.npz"),
             seq=torch.cat([ds_eth.obs_traj, ds_eth.pred_traj], dim=2).numpy(),
             seq_rel=torch.cat([ds_eth.obs_traj_rel, ds_eth.pred_traj_rel], dim=2).numpy(),
             num_peds=np.asarray([e - s for s, e in ds_eth.seq_start_end], dtype=np.int32),
             non_linear=ds_eth.non_linear_ped.numpy(), loss_mask=ds_eth.loss_mask.numpy(),
             max_peds_in_frame=np.int64(ds_eth.max_peds_in_frame),
             # graphs the reference built for the first windows (V_obs / A_obs / V_tr / A_tr)
             v_obs0=ds_eth.v_obs[0].numpy(), a_obs0=ds_eth.A_obs[0].numpy(),
             v_tr0=ds_eth.v_pred[0].numpy(), a_tr0=ds_eth.A_pred[0].numpy())

    # known-answer window counts of the other test splits with this repo's ingest
    counts = {}
    for name in ("eth", "hotel", "univ", "zara1", "zara2"):
        w = ingest[name][1] if name in ingest else my_data.load_windows(
            os.path.join(REF, "datasets", name, "test"), 8, 12, 1, with_non_linear=False)
        counts[name] = (len(w), int(w.num_peds.sum()), int(w.num_peds.max()))
        ingest.setdefault(name, (None, w))
        print("windows", name, counts[name])
    expected = {"eth": (70, 181, 5), "hotel": (301, 1053, 8), "univ": (947, 24334, 57),
                "zara1": (602, 2253, 14), "zara2": (921, 5833, 14)}
    assert counts == expected, counts

    # ---------------------------------------------------------------- adjacency (R1/R2)
    win_univ = ingest["univ"][1]
    picks = {}
    for v_want in (2, 3, 5):
        i = int(np.nonzero(win_eth.num_peds == v_want)[0][0])
        picks[v_want] = (win_eth, i)
    for v_want in (8,):
        w = ingest["hotel"][1]
        picks[v_want] = (w, int(np.nonzero(w.num_peds == v_want)[0][0]))
    for v_want in (17, 32, 57):
        i = int(np.nonzero(win_univ.num_peds == v_want)[0][0])
        picks[v_want] = (win_univ, i)
    adj = {}
    scenes = {}
    for v_want, (w, i) in picks.items():
        obs, pred, obs_rel, pred_rel = scene_from_windows(w, i)
        v_obs, a_obs = ref_utils.seq_to_graph(obs, obs_rel, True)
        v_tr, a_tr = ref_utils.seq_to_graph(pred, pred_rel, True)
        scenes[v_want] = (obs, pred, obs_rel, pred_rel, v_obs, a_obs, v_tr, a_tr)
        adj["rel_%d" % v_want] = obs_rel.numpy()
        adj["nodes_%d" % v_want] = v_obs.numpy()
        adj["lap_%d" % v_want] = a_obs.numpy()
        adj["predrel_%d" % v_want] = pred_rel.numpy()
        adj["prednodes_%d" % v_want] = v_tr.numpy()
        adj["predlap_%d" % v_want] = a_tr.numpy()
    # ties / zero-velocity case: equal displacements -> weight 0, standing peds, one mover
    tie = torch.zeros(4, 2, 8)
    tie[0, 0, 1:] = 0.25
    tie[1, 0, 1:] = 0.25            # same velocity as ped 0 -> edge weight 0
    tie[3, 1, 3:] = -0.5            # peds 2 and 3 standing until t=3
    v_t, a_t = ref_utils.seq_to_graph(torch.zeros(4, 2, 8), tie, True)
    adj["rel_tie"] = tie.numpy()
    adj["nodes_tie"] = v_t.numpy()
    adj["lap_tie"] = a_t.numpy()
    np.savez(os.path.join(out, "adj_cases.npz"), **adj)

    # ---------------------------------------------------------------- eval forward (R3-R5)
    m_eval = new_ref_model(weights=eth_sd)
    m_eval.eval()
    fwd = {}
    for v_want, sc in scenes.items():
        v_obs, a_obs = sc[4], sc[5]
        x = v_obs.unsqueeze(0).permute(0, 3, 1, 2)            # train.py:48
        with torch.no_grad():
            y, _ = m_eval(x, a_obs)
        fwd["vpred_%d" % v_want] = y.numpy()                   # (1,5,12,V)
        # module-level outputs for the boundary tests
        with torch.no_grad():
            g, _ = m_eval.st_gcns[0].gcn(x, a_obs)
            h, _ = m_eval.st_gcns[0](x, a_obs)
        fwd["gcn_%d" % v_want] = g.numpy()
        fwd["stgcn_%d" % v_want] = h.numpy()
    np.savez(os.path.join(out, "forward_eval.npz"), **fwd)

    # ---------------------------------------------------------------- train fwd+bwd (R3-R6)
    tr = {}
    for v_want in (3, 17, 57):
        sc = scenes[v_want]
        v_obs, a_obs, v_tr = sc[4], sc[5], sc[6]
        m = new_ref_model(seed=v_want)
        m.train()
        tr["init_%d" % v_want] = np.concatenate(
            [p.detach().numpy().ravel() for p in m.parameters()])
        for k, v in sd_to_np(m.state_dict()).items():
            tr["sd_%d/%s" % (v_want, k)] = v
        x = v_obs.unsqueeze(0).permute(0, 3, 1, 2)
        y, _ = m(x, a_obs)
        y.retain_grad()
        v_pred = y.permute(0, 2, 3, 1).squeeze(0)
        loss = ref_metrics.bivariate_loss(v_pred, v_tr)
        loss.backward()
        tr["vpred_%d" % v_want] = y.detach().numpy()
        tr["loss_%d" % v_want] = np.float64(loss.item())
        tr["dvpred_%d" % v_want] = y.grad.numpy()
        for name, p in m.named_parameters():
            tr["grad_%d/%s" % (v_want, name)] = (np.full(p.shape, np.nan, np.float32)
                                                 if p.grad is None else p.grad.numpy())
        for k, v in sd_to_np(m.state_dict()).items():
            if "running" in k or "num_batches" in k:
                tr["after_%d/%s" % (v_want, k)] = v
    np.savez(os.path.join(out, "train_fwd_bwd.npz"), **tr)

    # ---------------------------------------------------------------- loss cases (R6)
    g = torch.Generator().manual_seed(11)
    vp = torch.randn(12, 6, 5, generator=g)
    vt = torch.randn(12, 6, 2, generator=g)
    vp[0, 0, :] = torch.tensor([0.0, 0.0, -12.0, -12.0, 0.0])     # pdf underflows -> clamp active
    vt[0, 0, :] = torch.tensor([3.0, -3.0])
    vp[1, 1, :] = torch.tensor([0.1, 0.2, 0.3, -0.4, 4.0])        # rho -> ~1
    vp.requires_grad_(True)
    l = ref_metrics.bivariate_loss(vp, vt)
    l.backward()
    np.savez(os.path.join(out, "loss_cases.npz"), vpred=vp.detach().numpy(), vtrgt=vt.numpy(),
             loss=np.float64(l.item()), dvpred=vp.grad.numpy())

    # ---------------------------------------------------------------- reference train() loop (R9)
    import train as ref_train                                  # /root/reference/train.py
    n_sc, bs = 40, 16
    batches = []
    for i in range(n_sc):
        obs, pred, obs_rel, pred_rel = scene_from_windows(win_eth, i)
        item = [obs, pred, obs_rel, pred_rel, torch.zeros(obs.shape[0]), torch.ones(obs.shape[0], 20),
                ds_eth.v_obs[i], ds_eth.A_obs[i], ds_eth.v_pred[i], ds_eth.A_pred[i]]
        batches.append([t.unsqueeze(0) for t in item])         # DataLoader(batch_size=1)
    m = new_ref_model(seed=123)
    before = sd_to_np(m.state_dict())
    opt = torch.optim.SGD(m.parameters(), lr=0.01)
    targs = argparse.Namespace(batch_size=bs, clip_grad=None)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        ep_loss = ref_train.train(0, m, batches, opt, targs, torch.device("cpu"))
        m_val = new_ref_model(weights=m.state_dict())
        val_loss = ref_train.vald(0, m_val, batches, targs, torch.device("cpu"))
    grp = {"epoch_loss": np.float64(ep_loss), "val_loss": np.float64(val_loss),
           "n_scenes": np.int64(n_sc), "batch_size": np.int64(bs), "lr": np.float64(0.01)}
    for k, v in before.items():
        grp["before/" + k] = v
    for k, v in sd_to_np(m.state_dict()).items():
        grp["after/" + k] = v
    np.savez(os.path.join(out, "train_loop.npz"), **grp)

    # ---------------------------------------------------------------- test() sampling (R10)
    src = open(os.path.join(REF, "test.py")).read()
    head = src.split("paths = ['./checkpoint/*social-stgcnn*']")[0]
    ns = {}
    exec(compile(head, os.path.join(REF, "test.py"), "exec"), ns)       # defines test()
    ns["model"] = m_eval
    ev_batches = []
    for i in range(len(ds_eth)):
        ev_batches.append([t.unsqueeze(0) for t in ds_eth[i]])
    ns["loader_test"] = ev_batches
    torch.manual_seed(0)
    ade_, fde_, raw = ns["test"](KSTEPS=20)
    per_ade, per_fde = [], []
    for step in range(1, len(ev_batches) + 1):
        tgt = raw[step]["trgt"]
        n_obj = tgt.shape[1]
        for n in range(n_obj):
            a_ls, f_ls = [], []
            for pred in raw[step]["pred"]:
                a_ls.append(ref_metrics.ade([pred[:, n:n + 1, :]], [tgt[:, n:n + 1, :]], [1]))
                f_ls.append(ref_metrics.fde([pred[:, n:n + 1, :]], [tgt[:, n:n + 1, :]], [1]))
            per_ade.append(min(a_ls))
            per_fde.append(min(f_ls))
    assert abs(sum(per_ade) / len(per_ade) - ade_) < 1e-12
    with torch.no_grad():
        vp_all = []
        for b in ev_batches:
            y, _ = m_eval(b[6].permute(0, 3, 1, 2), b[7].squeeze())
            vp_all.append(y.permute(0, 2, 3, 1).squeeze(0).numpy().reshape(12, -1, 5))
    np.savez(os.path.join(out, "eval_ade_fde.npz"), ade=np.float64(ade_), fde=np.float64(fde_),
             per_ped_ade=np.asarray(per_ade), per_ped_fde=np.asarray(per_fde),
             vpred_cat=np.concatenate(vp_all, axis=1))
    print("eth test ADE/FDE seed0:", ade_, fde_)
    print("done ->", out)


if __name__ == "__main__":
    main()
