#!/usr/bin/env python
"""Full-size reference fixtures: BASELINE configs[2] (B=1024) and configs[3] (ActivityNet shape, B=128, M=1024).

Same method as capture_golden.py (the REFERENCE itself, imported unmodified, run on the CPU on the seeded inputs of
neighborretr_amd/synth.py; build container only), at the sizes the earlier fixtures only reached through properties.  The
reference's intermediates are GBs here ([1024,1024,24,12] f32 = 1.2 GB per temporary at configs[2]; [128,1024,64,64] =
2.1 GB per bank call at configs[3]), so everything runs under no_grad and the fixtures keep REDUCED forms of the big
matrices: row sums, column sums, the diagonal, a 64 x 64 corner -- plus everything that is small by nature (the five
losses, bank centralities, neighbour indices, `cols` of compute_metrics).

    c3_b1024      configs[2]: B=1024, Nt=24, Nv=12, M=512, K=20 -- the whole loss step (modeling.py:314-360)
    c4_b128_full  configs[3]: B=128, Nt=Nv=64, M=1024, K=20 -- the components the reference computes at these token counts
                  (its centrality term raises: until_module.py:321), incl. both full-size bank products as row means

Writes tests/golden/<name>.npz and appends the oracle-vs-reference deviations to tests/golden/CAPTURE_LOG.txt.
"""
import datetime
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import capture_golden as CG  # noqa: E402


def reduced(out, key, A, corner=64):
    A = A.detach().double()
    out[key + "_rowsum"] = A.sum(1).numpy()
    out[key + "_colsum"] = A.sum(0).numpy()
    out[key + "_diag"] = torch.diagonal(A).float().numpy()
    out[key + "_corner"] = A[:corner, :corner].float().numpy()


def neighbor_indices(nb_mask, K):
    """[B,K] int16: the K neighbour columns of every row, ascending (the mask itself is B x B)."""
    idx = torch.nonzero(nb_mask > 0.5)
    B = nb_mask.shape[0]
    assert idx.shape[0] == B * K
    return idx[:, 1].reshape(B, K).to(torch.int16).numpy()


def capture(name, NeighborRetr, RetrievalMetrics, seed, B, Nt, Nv, M, K, log):
    from neighborretr_amd import synth
    import nr_oracle as O
    hp = dict(synth.DEFAULT_HP, num_neighbors=K)
    P = synth.make_params(7)
    prob = synth.make_problem(seed, B, Nt, Nv, M)
    noise = synth.make_noise(seed, B, Nt, Nv)
    m = CG.build_reference_head(NeighborRetr, P, hp)
    Pt = {k: torch.from_numpy(v) for k, v in P.items()}
    x = CG.to_t(prob)
    nt = {k: torch.from_numpy(v) for k, v in noise.items()}
    out = dict(seed=seed, B=B, Nt=Nt, Nv=Nv, M=M, K=K, param_seed=7, blank_video=-1)
    dev = {}

    def cmp(key, ref, mine):
        r, q = ref.detach().double(), mine.detach().double()
        dev[key] = float((r - q).abs().max())

    tf, vf, tm, vm = x["text_feat"], x["video_feat"], x["text_mask"], x["video_mask"]
    with torch.no_grad():
        # a-4: the three call shapes
        S_ref, _ = m.local_level(tf, vf, tm, vm)
        S_o = O.local_level(tf, vf, tm, vm, Pt)[0]
        cmp("S", S_ref, S_o)
        reduced(out, "S", S_ref)
        met = RetrievalMetrics.compute_metrics(S_ref.numpy())
        out["cols"] = np.asarray(met["cols"], dtype=np.int32)
        out["metrics"] = np.array([met[k] for k in ("R1", "R5", "R10", "R50", "MR", "MeanR")], dtype=np.float64)
        bt2v_ref, _ = m.local_level(tf, x["mb_feat_v"], tm, x["mb_mask_v"])
        _, bv2t_ref = m.local_level(x["mb_feat_t"], vf, x["mb_mask_t"], vm)
        out["bank_c_t2v"] = bt2v_ref.mean(-1).numpy()
        out["bank_c_v2t"] = bv2t_ref.mean(-1).numpy()
        out["bank_t2v_corner"] = bt2v_ref[:64, :64].numpy()
        out["bank_v2t_corner"] = bv2t_ref[:64, :64].numpy()
        bt2v_o = O.local_level(tf, x["mb_feat_v"], tm, x["mb_mask_v"], Pt)[0]
        bv2t_o = O.local_level(x["mb_feat_t"], vf, x["mb_mask_t"], vm, Pt)[1]
        cmp("bank_c_t2v", bt2v_ref.mean(-1), bt2v_o.mean(-1))
        cmp("bank_c_v2t", bv2t_ref.mean(-1), bv2t_o.mean(-1))
        del bt2v_o, bv2t_o
        wl = m.text_weight_fc(tf).squeeze(2).masked_fill((1 - tm).to(torch.bool), float(-9e15))
        out["w_t"] = torch.softmax(wl, -1).numpy().astype(np.float32)
        wl = m.video_weight_fc(vf).squeeze(2).masked_fill((1 - vm).to(torch.bool), float(-9e15))
        out["w_v"] = torch.softmax(wl, -1).numpy().astype(np.float32)
        # a-7
        nal = m.neighbor_adjusting_loss
        nb_ref, _ = nal.create_neighbor_mask(S_ref, K)
        out["nb_idx"] = neighbor_indices(nb_ref, K)
        Ln_ref = m.compute_neighbor_loss(tf, vf, tm, vm, x["mb_feat_t"], x["mb_feat_v"], x["mb_mask_t"], x["mb_mask_v"],
                                         S_ref, S_ref.T, K, hp["temperature"])
        out["L_neighbor_direct"] = Ln_ref.numpy()
        cmp("L_neighbor_direct", Ln_ref, O.neighbor_loss(S_o, O.local_level(tf, x["mb_feat_v"], tm, x["mb_mask_v"], Pt)[0],
                                                         O.local_level(x["mb_feat_t"], vf, x["mb_mask_t"], vm, Pt)[1],
                                                         K, hp["temperature"]))
        # a-10 / a-8
        with CG.FeedRand(CG.noise_queue(noise)):
            gt_ref, gv_ref = m.merge_global_features(tf, vf, tm, vm)
        gt_o, gv_o = O.merge_global_features(tf, vf, tm, vm, Pt, nt, centre_ties="torch_topk")
        cmp("gt", gt_ref, gt_o)
        cmp("gv", gv_ref, gv_o)
        out["gt_rowsum"] = gt_ref.double().sum(-1).numpy()          # [B, G]
        out["gv_rowsum"] = gv_ref.double().sum(-1).numpy()
        out["gt_head"] = gt_ref[:16].numpy()
        out["gv_head"] = gv_ref[:16].numpy()
        G_ref, _ = m.global_level(gt_ref, gv_ref)
        cmp("G", G_ref, O.global_level(gt_o, gv_o, Pt)[0])
        reduced(out, "G", G_ref)
        url = m.uniform_regularization_loss
        tgt_r = url.sinkhorn_algorithm(G_ref, hp["beta"], 50)
        tgt_c = url.sinkhorn_algorithm(G_ref.T, hp["beta"], 50)
        reduced(out, "tgt_t2v", tgt_r)
        reduced(out, "tgt_v2t", tgt_c)
        cmp("tgt_t2v", tgt_r, O.sinkhorn_targets(G_ref, hp["beta"]))
        Lkl = (m.kl_loss(G_ref, S_ref) + m.kl_loss(G_ref.T, S_ref.T)) / 2
        out["L_kl_direct"] = Lkl.numpy()
        Lu = (url(G_ref, hp["temperature"], hp["beta"]) + url(G_ref.T, hp["temperature"], hp["beta"])) / 2
        out["L_uniform_direct"] = Lu.numpy()
        cmp("L_uniform_direct", Lu, O.uniform_loss(G_ref, hp["temperature"], hp["beta"]))
        if gt_ref.shape[1] == 1:
            wt_ref, wv_ref = m.compute_centrality_weights(tf, vf, gt_ref, gv_ref, hp["centrality_scale"])
            out["w_text"], out["w_video"] = wt_ref.numpy(), wv_ref.numpy()
            ls = torch.tensor(100.0)
            with CG.FeedRand(CG.noise_queue(noise)):
                losses = m._compute_losses(tf, vf, tm, vm, x["mb_feat_t"], x["mb_feat_v"], x["mb_mask_t"], x["mb_mask_v"],
                                           hp["centrality_scale"], hp["beta"], K, hp["temperature"], ls)
            out["losses"] = torch.stack(list(losses)).numpy()
            lo = O.compute_losses(tf, vf, tm, vm, x["mb_feat_t"], x["mb_feat_v"], x["mb_mask_t"], x["mb_mask_v"], Pt, hp, ls, nt)
            cmp("losses", torch.stack(list(losses)), torch.stack(list(lo)))
        else:
            try:
                wt_ref, _ = m.compute_centrality_weights(tf, vf, gt_ref, gv_ref, hp["centrality_scale"])
                m.centrality_weighting_loss(S_ref * 100.0, wt_ref)
                out["centrality_raises"] = 0
            except RuntimeError:
                out["centrality_raises"] = 1
    np.savez_compressed(os.path.join(CG.OUT, name + ".npz"), **out)
    size = os.path.getsize(os.path.join(CG.OUT, name + ".npz"))
    lines = [f"[{name}] B={B} Nt={Nt} Nv={Nv} M={M} K={K}: oracle-vs-reference max|diff| = {max(dev.values()):.3e}  ({size} bytes)"]
    lines += [f"    {k:20s} {v:.3e}" for k, v in dev.items()]
    if "losses" in out:
        lines.append("    reference losses     " + " ".join(f"{v:.6f}" for v in out["losses"]))
    for ln in lines:
        print(ln, flush=True)
        log.write(ln + "\n")


def main():
    if not os.path.isdir(CG.REF):
        print("reference checkout not present; nothing to capture")
        return 0
    torch.manual_seed(0)
    torch.set_num_threads(8)
    NeighborRetr, RetrievalMetrics = CG._import_reference()
    which = sys.argv[1:] or ["c3_b1024", "c4_b128_full"]
    with open(os.path.join(CG.OUT, "CAPTURE_LOG.txt"), "a") as log:
        log.write(f"{datetime.date.today()} capture_golden_large.py\n")
        if "c3_b1024" in which:
            capture("c3_b1024", NeighborRetr, RetrievalMetrics, seed=1003, B=1024, Nt=24, Nv=12, M=512, K=20, log=log)
        if "c4_b128_full" in which:
            capture("c4_b128_full", NeighborRetr, RetrievalMetrics, seed=3004, B=128, Nt=64, Nv=64, M=1024, K=20, log=log)
    return 0


if __name__ == "__main__":
    sys.exit(main())
