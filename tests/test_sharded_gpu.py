"""GPU, two processes (gloo) on one card: the loss sharded over ranks (row slabs of S, 1/W of the bank
products, all-gather of the centrality slices, all-reduce of the row terms) equals the replicated loss."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from neighborretr_amd import modeling, synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    B, Nt, Nv, M, K = 32, 24, 12, 64, 8
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
    m = m.to(dev).train()
    m.config.world_size = world
    p = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_problem(2024, B, Nt, Nv, M).items()}
    nz = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_noise(2024, B, Nt, Nv).items()}
    c = m.config
    args = (p["text_feat"], p["video_feat"], p["text_mask"].float(), p["video_mask"].float(), p["mb_feat_t"], p["mb_feat_v"],
            p["mb_mask_t"].float(), p["mb_mask_v"].float(), c.centrality_scale, c.beta, K, c.temperature,
            torch.tensor(100.0, device=dev))
    with torch.no_grad():
        m.shard_loss = False
        ref = torch.stack(m._compute_losses(*args, noise=nz)).cpu()
        m.shard_loss = True
        got = torch.stack(m._compute_losses(*args, noise=nz)).cpu()
        # sample-sharded clustering alone, on data where the batch-wide maximum distance matters: a sample with two valid
        # tokens (fewer than k = 3: its kNN density sees the fill value) on rank 0, the batch's farthest token pair on rank 1
        tf, vf, tm, vm = p["text_feat"].clone(), p["video_feat"].clone(), p["text_mask"].float().clone(), p["video_mask"].float().clone()
        tm[1, 2:] = 0
        vm[2, 2:] = 0
        tf[B - 1, 1] = -tf[B - 1, 0]                  # distances are taken between LayerNorm outputs: opposite tokens are
        vf[B - 2, 1] = -vf[B - 2, 0]                  # as far apart as tokens get
        ref_t, ref_v = m._merge_grouped(tf, vf, tm, vm, nz)
        rows = slice(rank * (B // world), (rank + 1) * (B // world))
        got_t, got_v = m._merge_sharded(tf, vf, tm, vm, nz, rank, world)
        shard_ok = bool(torch.equal(got_t, ref_t[rows]) and torch.equal(got_v, ref_v[rows]))
        all_t, all_v = m._gather_global(got_t, got_v, world)
        gather_ok = bool(torch.equal(all_t, ref_t) and torch.equal(all_v, ref_v))
        # the exchange itself: rank 0's own maximum is smaller than the batch's (the farthest pair sits on rank 1); what the
        # back kernel reads after the exchange is the batch-wide one
        import neighborretr_amd.cluster_fused as CF
        seen = {}

        def exchange(smax):
            seen["local"] = [float(x.max()) for x in smax]
            g = torch.stack([x.max() for x in smax])
            dist.all_reduce(g, op=dist.ReduceOp.MAX)
            for x, v in zip(smax, g):
                x[:1] = v
            seen["after"] = [float(x.max()) for x in smax]
        CF.ctm_stage_group([("text0", tf[rows].contiguous(), tm[rows].contiguous(), m.text_ctm0, m.text_block0, nz["t0"][rows].contiguous()),
                            ("video0", vf[rows].contiguous(), vm[rows].contiguous(), m.video_ctm0, m.video_block0,
                             nz["v0"][rows].contiguous())], m._ctm_cache, exchange=exchange)
        torch.cuda.synchronize()
    torch.save({"ref": ref, "got": got, "shard_ok": shard_ok, "gather_ok": gather_ok, "seen": seen},
               f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_loss_equals_replicated_loss(tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, 29611
    out = str(tmp_path / "res")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    res = [torch.load(f"{out}.{r}") for r in range(world)]
    for r in res:
        assert torch.isfinite(r["ref"]).all()
        assert torch.allclose(r["got"], r["ref"], rtol=2e-6, atol=2e-6), (r["got"], r["ref"])
    assert torch.equal(res[0]["got"], res[1]["got"])              # every rank holds the same losses
    # sample-sharded clustering == the rank's rows of the replicated clustering, bit for bit; rank 0 (short samples, but not
    # the batch's farthest pair) reads the batch-wide maximum distance after the exchange
    assert all(r["shard_ok"] and r["gather_ok"] for r in res)
    for k in (0, 1):                                               # text, video
        assert res[0]["seen"]["local"][k] < res[1]["seen"]["local"][k]
        assert res[0]["seen"]["after"][k] == res[1]["seen"]["after"][k] == res[1]["seen"]["local"][k]


def _train_worker(rank, world, port, out_path, precision):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import torch.distributed as dist
    from neighborretr_amd import modeling, synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    B, Nt, Nv, M, K = 32, 24, 12, 64, 8
    b = B // world
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K, world_size=world, local_rank=rank), precision=precision)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
    m = m.to(dev).train()
    with torch.no_grad():
        m.clip.logit_scale.fill_(float(np.log(100.0)))
    full = synth.make_problem(2024, B, Nt, Nv, M)
    sl = slice(rank * b, (rank + 1) * b)
    bank = {k: torch.from_numpy(full[k]).to(dev) for k in ("mb_feat_t", "mb_feat_v", "mb_mask_t", "mb_mask_v")}
    res = {}
    for mode in ("replicated", "sharded"):
        m.shard_loss = mode == "sharded"
        m.mb_feat_t, m.mb_feat_v = bank["mb_feat_t"].clone(), bank["mb_feat_v"].clone()
        m.mb_mask_t, m.mb_mask_v = bank["mb_mask_t"].clone(), bank["mb_mask_v"].clone()
        m.mb_ind = torch.arange(M, device=dev)
        m._rng_state = None
        torch.manual_seed(1234)                       # the same DPC-KNN noise stream in both runs
        tf = torch.from_numpy(full["text_feat"][sl]).to(dev).requires_grad_(True)
        vf = torch.from_numpy(full["video_feat"][sl]).to(dev).requires_grad_(True)
        m.zero_grad(set_to_none=True)
        losses = m(tf, torch.from_numpy(full["text_mask"][sl]).to(dev), vf, torch.from_numpy(full["video_mask"][sl]).to(dev),
                   torch.from_numpy(full["idx"][sl]).to(dev), 0)
        losses[0].backward()
        grads = {}
        for n, p in m.named_parameters():
            if p.grad is not None:
                g = p.grad.detach().clone()
                if mode == "sharded":                 # what DDP would do: the mean over the ranks
                    dist.all_reduce(g)
                    g /= world
                grads[n] = g.cpu()
        res[mode] = dict(losses=torch.stack([l.detach() for l in losses]).cpu(), g_text=tf.grad.cpu(), g_video=vf.grad.cpu(), params=grads)
    torch.save(res, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("precision,port", [("bf16x3", 29653), ("bf16", 29655)])
def test_sharded_training_loss_gradients_equal_replicated(tmp_path, precision, port):
    """(both precision plans: "bf16" is what `NeighborRetr()` trains in by default.)
    Training step, two ranks: losses, the gradient of this rank's features and the DDP-averaged parameter gradients of
    the sharded loss (neighborretr_amd.sharded: row slabs + differentiable collectives, reduce-scatter in the exchange
    step's backward) equal those of the reference's replicated loss."""
    import torch.multiprocessing as mp
    world = 2
    out = str(tmp_path / "res")
    mp.spawn(_train_worker, args=(world, port, out, precision), nprocs=world, join=True)
    fbar, pbar = 3e-3, 5e-3            # the same bars in both plans (measured: <= 6.3e-4 on the feature gradients)
    for r in range(world):
        res = torch.load(f"{out}.{r}", weights_only=False)
        rep, sh = res["replicated"], res["sharded"]
        assert torch.isfinite(rep["losses"]).all()
        assert torch.allclose(sh["losses"], rep["losses"], rtol=1e-4, atol=1e-4), (sh["losses"], rep["losses"])
        for k in ("g_text", "g_video"):
            scale = float(rep[k].abs().max())
            print(f"\n[sharded vs replicated, {precision}, rank {r}] {k}: max|d| / max|g| = {float((sh[k] - rep[k]).abs().max()) / scale:.2e}")
            assert float((sh[k] - rep[k]).abs().max()) < fbar * scale, (k, float((sh[k] - rep[k]).abs().max()), scale)
        assert set(sh["params"]) >= {n for n, g in rep["params"].items() if float(g.abs().max()) > 0}
        for n, g in rep["params"].items():
            scale = float(g.abs().max())
            if scale == 0:
                continue
            assert float((sh["params"][n] - g).abs().max()) < pbar * scale + 2e-6, (n, float((sh["params"][n] - g).abs().max()), scale)


def test_slab_row_losses_on_hip_match_the_torch_restatement():
    """SlabRowLossFn (nr_row_losses_fwd_slab / nr_row_losses_bwd_slab) against tests/slab_terms_torch.direction_terms -- the row terms restated
    in torch ops -- for one rank's slab: the partial losses and the gradients w.r.t. both S slabs, G, both bank centrality
    vectors, the centrality weights and the logit scale.  Single process (no collective involved)."""
    import numpy as np
    import slab_terms_torch
    from neighborretr_amd import ops, sharded, synth
    dev = "cuda"
    B, b, r0, K, T = 64, 16, 32, 8, 3.0
    hp = dict(num_neighbors=K, temperature=T, uniform_weight=1.0, neighbor_weight=0.7, kl_weight=1.3, beta=0.7)
    S = torch.from_numpy(synth.uniform(3, "slab_S", (B, B)).astype(np.float32) * 0.12).to(dev)
    G = torch.from_numpy(synth.normal(3, "slab_G", (B, B)).astype(np.float32) * 6).to(dev)
    c0 = torch.from_numpy(synth.uniform(3, "slab_c0", (B,)).astype(np.float32) * 0.1).to(dev)
    c1 = torch.from_numpy(synth.uniform(3, "slab_c1", (B,)).astype(np.float32) * 0.1).to(dev)
    w_t = torch.exp(torch.from_numpy(synth.normal(3, "slab_wt", (b,)).astype(np.float32) * 0.05)).to(dev)
    w_v = torch.exp(torch.from_numpy(synth.normal(3, "slab_wv", (b,)).astype(np.float32) * 0.05)).to(dev)
    tgt_r, tgt_c = ops.sinkhorn_targets(G, 0.7, 50)
    sl = slice(r0, r0 + b)

    def leaves():
        return [t.clone().requires_grad_(True) for t in (S[sl].contiguous(), S[:, sl].contiguous(), G, c0, c1, w_t, w_v,
                                                           torch.tensor([100.0], device=dev))]
    # torch restatement (normalised like the full loss)
    a = leaves()
    diag = torch.arange(r0, r0 + b, device=dev)
    ct, ut, nt, kt = slab_terms_torch.direction_terms(a[0], a[2][sl], tgt_r[sl], a[3], a[5], a[7].reshape(()), K, T, diag)
    cv, uv, nv, kv = slab_terms_torch.direction_terms(a[1].t(), a[2].t()[sl], tgt_c[sl], a[4], a[6], a[7].reshape(()), K, T, diag)
    cent, unif, neigh = ((x + y) / (2 * B) for x, y in ((ct, cv), (ut, uv), (nt, nv)))
    kl = (kt + kv) / (2 * B * B)
    ref = torch.stack((cent + unif * hp["uniform_weight"] + neigh * hp["neighbor_weight"] + kl * hp["kl_weight"], cent, unif, neigh, kl))
    wts = torch.tensor([1.0, 0.3, -0.2, 0.5, 0.1], device=dev)
    (ref * wts).sum().backward()
    # HIP
    h = leaves()
    wt_full = torch.zeros(B, device=dev).index_add(0, diag, h[5])
    wv_full = torch.zeros(B, device=dev).index_add(0, diag, h[6])
    got = sharded.SlabRowLossFn.apply(h[0], h[1], h[2], tgt_r, tgt_c, h[3], h[4], wt_full, wv_full, h[7], hp, r0)
    (got * wts).sum().backward()
    assert torch.allclose(got, ref, rtol=2e-5, atol=1e-6), (got, ref)
    for name, x, y in zip(("S_rows", "S_cols", "G", "c0", "c1", "w_text", "w_video", "logit_scale"), h, a):
        scale = float(y.grad.abs().max())
        err = float((x.grad - y.grad).abs().max())
        assert err < 2e-4 * scale + 1e-9, (name, err, scale)


def _interleaved_worker(rank, world, port, out_path, overlap=False, M=64):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from neighborretr_amd import modeling, synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    B, Nt, Nv, K, steps = 32, 24, 12, 8, 6
    b = B // world
    torch.manual_seed(11)                           # the noise stream's seed: the same on every rank, as in the entry point

    def build():
        m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K))
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
        m = m.to(dev).train()
        p = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_problem(77, B, Nt, Nv, M).items()}
        m.mb_feat_t, m.mb_feat_v, m.mb_mask_t, m.mb_mask_v = p["mb_feat_t"], p["mb_feat_v"], p["mb_mask_t"], p["mb_mask_v"]
        m.mb_ind = torch.arange(M, device=dev)
        return m
    batches = [{k: torch.from_numpy(v).to(dev) for k, v in synth.make_problem(500 + s, B, Nt, Nv, M).items()} for s in range(steps)]
    for s, p in enumerate(batches):
        p["idx"] = torch.arange(1000 * s, 1000 * s + B, device=dev)
    # the single-rank run: every step's losses, the bank at the end
    m1 = build()
    ref = []
    with torch.no_grad():
        for p in batches:
            ref.append(torch.stack(m1(p["text_feat"], p["text_mask"], p["video_feat"], p["video_mask"], p["idx"], 0)).cpu())
    # the step-interleaved job on the same stream of batches
    m = build()
    m.config.world_size, m.config.local_rank = world, rank
    m.interleave_steps = True
    m.interleave_overlap = overlap          # the owner's loss on a second stream, from a copy of the bank, beside the next steps
    sl = slice(rank * b, (rank + 1) * b)
    mine = {}
    with torch.no_grad():
        for s, p in enumerate(batches):
            out = m(p["text_feat"][sl].contiguous(), p["text_mask"][sl].contiguous(), p["video_feat"][sl].contiguous(),
                    p["video_mask"][sl].contiguous(), p["idx"][sl].contiguous(), 0)
            assert (out is not None) == (s % world == rank)
            if out is not None:
                mine[s] = out                           # (overlapped: still being computed -- read after the loop)
        m.wait_owned_loss()
        mine = {s: torch.stack(out).cpu() for s, out in mine.items()}
    assert (m._owned is not None) == overlap
    # the prepared shadow of the bank (normalised bf16 pairs + norms) is kept in step with the ring on BOTH kinds of step
    sh, sh1 = m._mb_shadow, m1._mb_shadow
    had_shadow = sh is not None and sh1 is not None
    shadow_same = had_shadow and all(torch.equal(getattr(a, f), getattr(b_, f)) for a, b_ in zip(sh, sh1) for f in ("hi", "lo", "norm"))
    heads = tuple(int(x._mb_head_dev.item()) if x._mb_head_dev is not None else x._mb_head for x in (m, m1))
    bank_same = all(torch.equal(getattr(m, k), getattr(m1, k)) for k in ("mb_ind", "mb_feat_t", "mb_feat_v", "mb_mask_t", "mb_mask_v"))
    torch.save({"ref": ref, "mine": mine, "bank_same": bank_same, "had_shadow": had_shadow, "shadow_same": shadow_same, "heads": heads},
               f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("M", [64, 16], ids=["bank-of-64", "bank-of-16"])
@pytest.mark.parametrize("overlap", [False, True], ids=["loss-in-front", "loss-beside"])
def test_step_interleaved_job_equals_the_single_rank_run_bit_for_bit(tmp_path, overlap, M):
    """model.interleave_steps on two ranks (gloo, one card): every step is gathered and pushed on both ranks, its loss evaluated
    on rank (step mod 2).  The losses of every step and the memory bank after the last one are those of the single-rank run on
    the same stream of batches -- identical bits: same kernels, same ring, same noise stream (the reference's semantics:
    modeling.py:274-312, every step sees the bank left by the steps before it).  overlap: model.interleave_overlap -- the owner
    copies the bank's prepared shadow, pushes the batch at once and evaluates its loss from the copy on a second stream while
    the following steps' exchanges and pushes already run (modeling.OwnedSlot); the same bits again.  Bank of 16: every gathered
    batch (32) replaces the bank (modeling.py:244-249) -- the same launches with the ring head at 0."""
    import torch.multiprocessing as mp
    world, port = 2, 29641 + int(overlap) + (2 if M == 16 else 0)
    out = str(tmp_path / "res")
    mp.spawn(_interleaved_worker, args=(world, port, out, overlap, M), nprocs=world, join=True)
    res = [torch.load(f"{out}.{r}") for r in range(world)]
    steps = len(res[0]["ref"])
    for s in range(steps):
        owner = res[s % world]
        assert s in owner["mine"] and s not in res[(s + 1) % world]["mine"]
        assert torch.isfinite(owner["ref"][s]).all()
        assert torch.equal(owner["mine"][s], owner["ref"][s]), (s, owner["mine"][s], owner["ref"][s])
    assert all(r["bank_same"] and r["had_shadow"] and r["shadow_same"] and r["heads"][0] == r["heads"][1] for r in res), res
    assert not torch.equal(res[0]["ref"][0], res[0]["ref"][2])          # the steps differ (new batch, moved bank)
