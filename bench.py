#!/usr/bin/env python
"""Benchmark of the NeighborRetr similarity / loss head on MI355X.

    python bench.py --gpus N --steps K --warmup W

One STEP = one pass of the hot path over one synthetic batch already resident in HBM:
    [N>1: RCCL all-gather of the per-rank feature/mask shards]  ->  token clustering (global
    tokens)  ->  prepare / scorer / fused local_level x3 / Sinkhorn / row losses  ->  memory-bank
    FIFO push,   i.e. `NeighborRetr.forward` in training mode minus the encoders, loss-only
    forward (BASELINE.json configs[0] "loss-only forward").
Workload = BASELINE.json configs[1]: global B=128, d=512, Nt=24, Nv=12, M=512, K=20 (MSR-VTT shape); `--config 2|3` run
configs[2] / [3] on ONE GPU as extra lines.  The step is replayed from HIP graphs.
For N>1 the GLOBAL batch stays 128 (the metric is quoted at global B=128; `scaling` "strong"): each rank holds b=128/N samples of
every step.  Three forms of the job:
  default        step-interleaved: every step is gathered (one packed RCCL all-gather) and pushed into the bank replica on EVERY
                 rank; its LOSS is evaluated on rank (step mod N) with the single-rank kernels -- loss-only steps depend on each
                 other through the bank alone.  Losses and bank are bit-identical to the single-rank run.  The owner's loss
                 can run BESIDE the following steps (a copy of the bank, a second graph on a second stream:
                 neighborretr_amd/interleave.py): that form is built on a few draws of fresh streams next to the serial one,
                 every form is validated against the eager steps, a few rounds of each are timed and the fastest is kept --
                 by all ranks together (`config.owner_loss`, `config.form_probes`; `--no_overlap`: the serial form only).
  --sync_step    the synchronous sharded step: every rank takes part in every loss (five collectives per step).
  --replicated_loss   the reference's semantics (modeling.py:274-298): every rank evaluates every loss.
A step that contains collectives is replayed as ONE graph with the RCCL collectives inside, or as the graphs of its rank-local
segments with the collectives eager between them, or eagerly -- the ranks decide together (comm.CollectiveCapture); the line says
which (`config.step_form`).  The JSON line also carries:
  roofline     the fused local_level kernel (the dominant kernel): algorithmic flops per launch
               (F_sim / 3 launches) / its average duration, timed live
               with HIP events on the launch stream, against the dense bf16 MFMA peak;
  cpu_baseline the CPU oracle (a port of the reference algorithm, oracle/nr_oracle.py) timed on this
               host's cores on the same workload (N=1, rank 0 only; bounded sample);
  parity       one un-timed step against the reference fixture of this workload;
  rank_local   (--emulate_world) what one rank of a W-rank job does, emulated on this GPU.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
CFG = dict(B=128, Nt=24, Nv=12, M=512, K=20, d=512)
# BASELINE.json configs[k]: [1] is the headline workload (the default, what `value` is quoted on); [2] and [3] are the two
# MFMA-bound shapes, run on ONE GPU as extra lines (`--config 2|3`; profiles/r04_bench_c2.json, _c3.json)
CFGS = {1: dict(B=128, Nt=24, Nv=12, M=512, K=20, d=512, name="BASELINE configs[1]: B=128 d=512 Nt=24 Nv=12 M=512 K=20 (MSR-VTT shape)"),
        2: dict(B=1024, Nt=24, Nv=12, M=512, K=20, d=512, name="BASELINE configs[2] on ONE GPU: global B=1024 d=512 Nt=24 Nv=12 M=512 K=20"),
        3: dict(B=128, Nt=64, Nv=64, M=1024, K=20, d=512,
                name="BASELINE configs[3] on ONE GPU: ActivityNet shape B=128 d=512 Nt=64 Nv=64 M=1024 K=20, 3 / 6 global tokens per "
                     "sample (centrality_multi_token='mean': the reference raises at this shape, until_module.py:321)")}


def algorithmic_flops(B, Nt, Nv, M, d=512, H=1024):
    f_sim = 2 * d * ((B * Nt) * (B * Nv) + (B * Nt) * (M * Nv) + (M * Nt) * (B * Nv))
    f_mlp = (2 * d * H + 2 * H) * (B + M) * (Nt + Nv)
    return f_sim, f_mlp


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16x3", "bf16_all"])
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3],
                    help="BASELINE.json configs[k]: 1 = the headline workload (default); 2 (global B=1024) and 3 (ActivityNet token counts, "
                         "M=1024) run the same step on ONE GPU as extra lines -- no CPU baseline, no reference fixture at those sizes")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--unroll", type=int, default=0,
                    help="consecutive steps captured per HIP graph, every dependency between them kept (the bank push of step k before "
                         "the bank products of step k+1): one replay then issues U steps and the ~10 us between two replays is paid "
                         "once per U steps (round 3: 312 / 306 / 304 us per step at U = 1 / 2 / 4).  A remainder of K mod U steps is "
                         "replayed step by step.  0 (default) = 20, whatever --steps is (fewer only when fewer steps are timed): a graph's end costs ~90 us (the "
                         "last step's tail runs alone), 3570 / 3640 / 3670 steps/s at U = 10 / 20 / 40.  "
                         "N > 1: one graph per step (see --round_graph).  1 = one step per graph")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N>1 (nccl = RCCL; gloo only to rehearse ranks that share one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync_step", "--shard_loss", dest="sync_step", action="store_true",
                    help="N>1: the SYNCHRONOUS sharded step instead of the step-interleaved default -- every rank takes part in every "
                         "step's loss: similarity / bank / clustering work sharded over the ranks (head.head_forward_sharded), five "
                         "collectives per step.  This is what a training step needs; at B=128 it is a latency chain that sharding does "
                         "not shorten (profiles/r04_rank_local.txt)")
    ap.add_argument("--replicated_loss", action="store_true",
                    help="N>1: the reference's form instead (modeling.py:274-298): after the all-gather every rank evaluates the whole "
                         "loss; the exchange step stays eager, the loss is replayed from a HIP graph")
    ap.add_argument("--backward", action="store_true", help="also time forward+backward (reported as extra fields)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="N=1: the U steps of an unrolled graph strictly one after the other (A/B switch).  Default: step k+1 starts "
                         "as soon as step k has moved the ring head and pushed its batch, i.e. beside step k's Sinkhorn solve and row "
                         "losses (modeling.StepPipeline); the graph is held against U single-step replays before it is used")
    ap.add_argument("--round_graph", action="store_true",
                    help="step-interleaved job, A/B switch: a whole round (this rank's own step and the N - 1 others) as ONE graph instead "
                         "of one graph per step.  Emulated, it is SLOWER (660 vs 620 us per round at N = 8, profiles/r04_rank_local.txt): "
                         "inside a multi-stream graph every node costs more than in the single-stream graph of a non-owned step")
    ap.add_argument("--decouple_push", action="store_true",
                    help="pipelined steps, A/B switch: the next step's prologue does not wait for this step's bank push (per-step copy "
                         "of the ring head; only the bank chains wait).  Slower at every config (0.355 vs 0.293 ms per step at configs[1]; 454-510 vs 527 "
                         "steps/s at configs[3]) and NOT protected like the default order: steps then run so far ahead of each other that a "
                         "buffer freed on the origin stream can be handed out while a tail still reads it -- at configs[3] the graph failed "
                         "its check against single-step replays once in two runs (the bench then takes the sequential form)")
    ap.add_argument("--emulate_world", type=int, nargs="*", default=None, metavar="W",
                    help="N=1 only, extra field `rank_local` (never the headline): what ONE rank of the sharded step does at these world "
                         "sizes (default 2 4 8), emulated on this GPU -- its own messages through a 1-rank RCCL communicator, the peers' "
                         "parts pre-filled (tools/rank_local_times.py; no wire time)")
    ap.add_argument("--overlap", action="store_true", help="(accepted for older command lines: the overlapped owned step is always tried now)")
    ap.add_argument("--overlap_tries", type=int, default=4,
                    help="step-interleaved job: how many times the overlapped owned step is built (on fresh streams) before the fastest "
                         "validated form -- overlapped or serial -- is kept")
    ap.add_argument("--no_overlap", action="store_true",
                    help="step-interleaved job, A/B switch: the owner evaluates its loss IN FRONT of the following steps (round 4's first "
                         "form) instead of beside them from a copy of the bank (neighborretr_amd.interleave)")
    ap.add_argument("--fail_whole_capture", action="store_true",
                    help="rehearsal switch: rank 0 pretends its whole-step capture failed, so that every rank takes the segmented-graph "
                         "form together (tests the collective fallback decision)")
    ap.add_argument("--no_kernel_profile", action="store_true",
                    help="N=1: skip the two child runs under rocprofv3 (the step loop itself, and the similarity launches alone) that "
                         "`roofline.frac_in_step`, `roofline.frac_alone` and `roofline.kernels` are read from")
    ap.add_argument("--no_sync_probe", action="store_true",
                    help="N>1, step-interleaved default: skip the second timed field `sync_step` (the synchronous sharded step, SURVEY 8e)")
    ap.add_argument("--_child", action="store_true", help=argparse.SUPPRESS)       # a profiled child: the step loop only
    ap.add_argument("--e2e", action="store_true",
                    help="also time BASELINE configs[4] on this GPU: ViT-B/32 towers + temporal transformer (stock PyTorch-ROCm, "
                         "random init, bf16 autocast) feeding the HIP head from synthetic pixels, forward and forward+backward "
                         "(extra field `e2e`; never the headline value)")
    args = ap.parse_args()
    CFG.update(CFGS[args.config])
    if args.unroll <= 0:
        # a FIXED 20 steps per graph whatever --steps asks for (fewer only when fewer are timed): K // 20 replays of it, the K % 20
        # steps left over as single-step replays -- the graph is not sized to the caller's command line
        args.unroll = min(max(args.steps, 1), 20)
    # (a batch as large as the bank -- configs[2] on one GPU -- replaces it IN PLACE, modeling.update_memory_bank: consecutive
    # steps of one graph see each other's bank there too; the unrolled graph is held against single-step replays like any other)
    return args


def build_model(precision, device):
    from neighborretr_amd import modeling, synth
    over = dict(centrality_multi_token="mean") if CFG["Nt"] > 24 else {}
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=CFG["K"], **over), precision=precision)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
    m = m.to(device).train()
    with torch.no_grad():
        m.clip.logit_scale.fill_(float(np.log(100.0)))
    if os.environ.get("NR_BANK_EARLY"):             # developer hook: how many bank chains start beside the clustering (0..2)
        m.bank_early = int(os.environ["NR_BANK_EARLY"])
    if os.environ.get("NR_CAPTURE_ORDER"):          # developer hook (tools/ab_tail.sh): "7,5;7,inf"
        m.capture_order = tuple(tuple((1 << 30) if x == "inf" else int(x) for x in t.split(",")) for t in os.environ["NR_CAPTURE_ORDER"].split(";"))
    return m


def host_cores():
    """(logical CPUs this process may run on, physical cores among them) from the affinity mask and /proc/cpuinfo."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    phys = set()
    try:
        cpu = pid = cid = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                cpu, pid, cid = int(v), None, None
            elif k == "physical id":
                pid = int(v)
            elif k == "core id":
                cid = int(v)
            elif not k and cpu is not None:
                if cpu in allowed and pid is not None and cid is not None:
                    phys.add((pid, cid))
                cpu = None
        if cpu is not None and cpu in allowed and pid is not None and cid is not None:
            phys.add((pid, cid))
    except OSError:
        pass
    return len(allowed), (len(phys) or len(allowed))


def cpu_quota():
    """CPUs this container may use at once (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown: a GPU
    box gives each GPU a share of the host's cores, and threads beyond it only fight over the same quota."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, q // p)
    except (OSError, ValueError):
        pass
    return None


def cpu_baseline(samples=7):
    """The CPU port of the reference algorithm (oracle/nr_oracle.py, pinned against the reference's own outputs) on the
    same workload, timed on this host: forward at 1 thread and at one thread per physical core (median of `samples`
    after 2 warm-ups each), forward+backward at all cores (median of 3).  `value` = the all-core forward rate."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nr_oracle as O
    from neighborretr_amd import synth
    logical, physical = host_cores()
    quota = cpu_quota()
    threads = min(physical, quota) if quota else physical
    c = CFG
    prob = {k: torch.from_numpy(v) for k, v in synth.make_problem(1002, c["B"], c["Nt"], c["Nv"], c["M"]).items()}
    P = {k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}
    nz = {k: torch.from_numpy(v) for k, v in synth.make_noise(1002, c["B"], c["Nt"], c["Nv"]).items()}
    hp = dict(synth.DEFAULT_HP, num_neighbors=c["K"])

    def one(grad=False):
        ls = torch.tensor(100.0, requires_grad=grad)
        tf, vf = prob["text_feat"].clone().requires_grad_(grad), prob["video_feat"].clone().requires_grad_(grad)
        Pg = {k: v.clone().requires_grad_(grad) for k, v in P.items()} if grad else P
        with torch.set_grad_enabled(grad):
            out = O.compute_losses(tf, vf, prob["text_mask"], prob["video_mask"], prob["mb_feat_t"], prob["mb_feat_v"],
                                   prob["mb_mask_t"], prob["mb_mask_v"], Pg, hp, ls, nz)
            if grad:
                out[0].backward()
        return out

    def timed(threads, n, grad=False, warm=2):
        torch.set_num_threads(threads)
        for _ in range(warm):
            one(grad)
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            one(grad)
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))
    t_all = timed(threads, samples)
    t_one = timed(1, samples)
    t_fb = timed(threads, 3, grad=True, warm=1)
    return {"value": round(1.0 / t_all, 3), "unit": "steps/s", "cores": threads, "kind": "port",
            "physical_cores": physical, "logical_cpus": logical, "cpu_quota": quota,
            "sample": f"configs[1] (B=128,Nt=24,Nv=12,M=512,K=20), torch CPU ops: forward x{samples} at {threads} threads "
                      f"(host: {physical} physical cores, this container's CPU quota: {quota or 'none'}) and x{samples} at 1 "
                      f"thread, forward+backward x3 at {threads} threads; medians after warm-ups",
            "ms_per_step": round(t_all * 1e3, 2),
            "one_thread": {"value": round(1.0 / t_one, 3), "ms_per_step": round(t_one * 1e3, 2), "cores": 1},
            "fwd_bwd": {"value": round(1.0 / t_fb, 3), "ms_per_step": round(t_fb * 1e3, 2), "cores": threads}}


def parity_gates(model, dev):
    """the gates reported beside every timing.  One extra UN-TIMED step on the inputs and the DPC-KNN noise
    of the reference fixture tests/golden/c2_b128.npz (outputs of the unmodified reference on exactly this workload):
    max|dS| of the batch x batch similarity in the step's precision plan, |dL| of the five losses, and whether the
    retrieval ranks (`cols` of RetrievalMetrics.compute_metrics, metrics.py:58-66) are identical on the rank-exact
    path.  The reference's cols are recomputed here from ITS similarity matrix with its own three numpy lines."""
    from neighborretr_amd import head, synth
    from neighborretr_amd.metrics import RetrievalMetrics
    c = CFG
    if (c["B"], c["Nt"]) != (128, 24):
        return parity_gates_full_size(model, dev)        # --config 2 / 3: the reduced full-size fixtures
    path = os.path.join(ROOT, "tests", "golden", "c2_b128.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    if (int(g["B"]), int(g["Nt"]), int(g["Nv"]), int(g["M"]), int(g["K"])) != (c["B"], c["Nt"], c["Nv"], c["M"], c["K"]):
        return None
    assert (int(g["B"]), int(g["Nt"]), int(g["Nv"]), int(g["M"]), int(g["K"])) == (c["B"], c["Nt"], c["Nv"], c["M"], c["K"])
    prob = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_problem(int(g["seed"]), c["B"], c["Nt"], c["Nv"], c["M"]).items()}
    nz = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_noise(int(g["seed"]), c["B"], c["Nt"], c["Nv"]).items()}
    cfg = model.config
    with torch.no_grad():
        losses = torch.stack(model._compute_losses(
            prob["text_feat"], prob["video_feat"], prob["text_mask"], prob["video_mask"], prob["mb_feat_t"], prob["mb_feat_v"],
            prob["mb_mask_t"], prob["mb_mask_v"], cfg.centrality_scale, cfg.beta, c["K"], cfg.temperature,
            torch.tensor(100.0, device=dev), noise=nz)).cpu().numpy()
        sw_t, sw_v = model.scorer_weights("text_weight_fc"), model.scorer_weights("video_weight_fc")
        p_bb = head.precision_plan(model._prec())[0]
        S_plan = head.similarity_matrix(prob["text_feat"], prob["video_feat"], prob["text_mask"].float(),
                                        prob["video_mask"].float(), sw_t, sw_v, p_bb)
        S_exact, _ = model.get_similarity_logits(prob["text_feat"], prob["video_feat"], prob["text_mask"], prob["video_mask"])
        mine = RetrievalMetrics.compute_metrics(S_exact)
    ref_S = g["S"]
    sx = np.sort(-ref_S, axis=1)
    ref_cols = np.where(sx - np.diag(-ref_S)[:, None] == 0)[1]
    dL = np.abs(losses - g["losses"])
    return {"fixture": "tests/golden/c2_b128.npz (reference outputs on this workload)",
            "max_dS": float(np.abs(S_plan.cpu().numpy() - ref_S).max()),
            "max_dS_rank_exact_path": float(np.abs(S_exact.cpu().numpy() - ref_S).max()),
            "dL": [float(f"{x:.3e}") for x in dL], "dL_order": ["total", "centrality", "uniform", "neighbor", "kl"],
            "losses": [round(float(x), 5) for x in losses], "ref_losses": [round(float(x), 5) for x in g["losses"]],
            "cols_identical": bool(np.array_equal(np.asarray(mine["cols"]), ref_cols)),
            "R1": mine["R1"], "tol_losses": 1e-3, "pass": bool(dL.max() < 1e-3 and np.array_equal(np.asarray(mine["cols"]), ref_cols))}


def parity_gates_full_size(model, dev):
    """The same gates at configs[2] / configs[3]: one extra UN-TIMED step on the inputs of the full-size reference fixtures
    (tests/golden/c3_b1024.npz: B=1024; c4_b128_full.npz: ActivityNet token counts, M=1024 -- outputs of the unmodified
    reference, oracle/capture_golden_large.py).  The fixtures hold reduced forms of the big matrices (row / column sums,
    diagonal, a 64 x 64 corner) plus the losses and `cols`.  At configs[3] the reference's centrality term raises
    (until_module.py:321): the three terms it does evaluate are compared, the other two are reported as unpinned."""
    from neighborretr_amd import head, synth
    from neighborretr_amd.metrics import RetrievalMetrics
    c = CFG
    name = "c3_b1024" if c["Nt"] == 24 else "c4_b128_full"
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    if (int(g["B"]), int(g["Nt"]), int(g["Nv"]), int(g["M"]), int(g["K"])) != (c["B"], c["Nt"], c["Nv"], c["M"], c["K"]):
        return None
    prob = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_problem(int(g["seed"]), c["B"], c["Nt"], c["Nv"], c["M"]).items()}
    nz = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_noise(int(g["seed"]), c["B"], c["Nt"], c["Nv"]).items()}
    cfg = model.config
    with torch.no_grad():
        losses = torch.stack(model._compute_losses(
            prob["text_feat"], prob["video_feat"], prob["text_mask"], prob["video_mask"], prob["mb_feat_t"], prob["mb_feat_v"],
            prob["mb_mask_t"], prob["mb_mask_v"], cfg.centrality_scale, cfg.beta, c["K"], cfg.temperature,
            torch.tensor(100.0, device=dev), noise=nz)).cpu().numpy()
        sw_t, sw_v = model.scorer_weights("text_weight_fc"), model.scorer_weights("video_weight_fc")
        p_bb = head.precision_plan(model._prec())[0]
        S_plan = head.similarity_matrix(prob["text_feat"], prob["video_feat"], prob["text_mask"].float(),
                                        prob["video_mask"].float(), sw_t, sw_v, p_bb)
        S_exact, _ = model.get_similarity_logits(prob["text_feat"], prob["video_feat"], prob["text_mask"], prob["video_mask"])
        mine = RetrievalMetrics.compute_metrics(S_exact)

    def corner_diag(S):
        S = S.double().cpu()
        return max(float((S[:64, :64] - torch.from_numpy(g["S_corner"]).double()).abs().max()),
                   float((torch.diagonal(S) - torch.from_numpy(g["S_diag"]).double()).abs().max()))

    def sums(S):
        S = S.double().cpu()
        return max(float((S.sum(1) - torch.from_numpy(g["S_rowsum"])).abs().max()), float((S.sum(0) - torch.from_numpy(g["S_colsum"])).abs().max()))
    cols_same = bool(np.array_equal(np.asarray(mine["cols"]), g["cols"]))
    out = {"fixture": f"tests/golden/{name}.npz (reference outputs on this workload; reduced forms of the {c['B']} x {c['B']} matrices)",
           "max_dS_corner_and_diagonal": corner_diag(S_plan), "max_dS_corner_and_diagonal_rank_exact_path": corner_diag(S_exact),
           "max_d_rowcol_sums_rank_exact_path": sums(S_exact), "cols_identical": cols_same, "R1": mine["R1"], "tol_losses": 1e-3,
           "losses": [round(float(x), 5) for x in losses], "dL_order": ["total", "centrality", "uniform", "neighbor", "kl"]}
    if "losses" in g:
        dL = np.abs(losses - g["losses"])
        out.update(dL=[float(f"{x:.3e}") for x in dL], ref_losses=[round(float(x), 5) for x in g["losses"]])
    else:
        dL = np.abs(losses[2:] - np.array([float(g["L_uniform_direct"]), float(g["L_neighbor_direct"]), float(g["L_kl_direct"])]))
        out.update(dL=[None, None] + [float(f"{x:.3e}") for x in dL],
                   ref_losses=[None, None] + [round(float(g[k]), 5) for k in ("L_uniform_direct", "L_neighbor_direct", "L_kl_direct")],
                   unpinned="total and centrality: the reference raises on 3 / 6 global tokens per sample (until_module.py:321); the "
                            "step runs config.centrality_multi_token='mean'")
    out["pass"] = bool(dL.max() < 1e-3 and cols_same)
    return out


def e2e_bench(dev, B=128, steps=5):
    """configs[4] at one GPU: pixels [B, 12, 3, 224, 224] + token ids [B, 24] -> encoders -> head -> losses (-> backward)."""
    from neighborretr_amd import modeling, synth
    from neighborretr_amd.encoders import synthetic_text_ids
    c = CFG
    torch.manual_seed(0)
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=c["K"], num_hidden_layers=4), with_encoders=True).to(dev).train()
    _, _, tm, vm = synth.make_samples(5005, "e2e", B, c["Nt"], c["Nv"], d=8)
    tm, vm = torch.from_numpy(tm).to(dev), torch.from_numpy(vm).to(dev)
    ids = synthetic_text_ids(tm.cpu(), seed=5).to(dev)
    video = torch.randn((B, c["Nv"], 3, 224, 224), device=dev)
    idx = torch.arange(B, device=dev)
    with torch.no_grad():                                   # the bank: M encoded samples (memory_bank.py:80-229)
        tf, vf = m.get_text_video_feat(ids, tm, video, vm)
        reps = c["M"] // B
        m.mb_feat_t, m.mb_feat_v = tf.repeat(reps, 1, 1).contiguous(), vf.repeat(reps, 1, 1).contiguous()
        m.mb_mask_t, m.mb_mask_v = tm.repeat(reps, 1).float(), vm.repeat(reps, 1).float()
        m.mb_ind = torch.arange(c["M"], device=dev)

    def fwd():
        with torch.no_grad():
            return m(ids, tm, video, vm, idx, 0)

    def fwd_bwd():
        m.zero_grad(set_to_none=True)
        m(ids, tm, video, vm, idx, 0)[0].backward()

    out = {}
    for name, fn in (("forward", fwd), ("forward_backward", fwd_bwd)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        out[name + "_ms"] = round((time.perf_counter() - t0) / steps * 1e3, 2)
    out["workload"] = (f"BASELINE configs[4] on ONE GPU: B={B}, 12 frames x 224x224, 24 tokens, bank {c['M']}; ViT-B/32 + text tower + "
                       "4-layer temporal transformer (stock PyTorch-ROCm modules, random init, bf16 autocast, SDPA) -> HIP head")
    out["samples_per_s_forward_backward"] = round(B / (out["forward_backward_ms"] * 1e-3), 1)
    out["params_millions"] = round(sum(p.numel() for p in m.parameters()) / 1e6, 1)
    del m
    torch.cuda.empty_cache()
    return out


# Per-kernel algorithmic work of ONE step at configs[1] (F_sim, F_mlp; the clustering's GEMMs as dense products),
# keyed by a substring of the kernel's name as rocprofv3 prints it: (role, launches per step, algorithmic flops per step, bound).
def kernel_table(c):
    B, Nt, Nv, M, d, H = c["B"], c["Nt"], c["Nv"], c["M"], c["d"], 1024
    t0, t1 = -(-Nt // 6), 1
    v0, v1 = -(-Nv // 4), 1
    r0, r1 = B * (Nt + Nv), B * (t0 + v0)                     # token rows entering clustering stage 0 / stage 1
    q0, q1 = B * (t0 + v0), B * (t1 + v1)                     # merged rows (queries) of stage 0 / stage 1
    f_sim, f_mlp = algorithmic_flops(B, Nt, Nv, M)
    return [
        ("nr_sim_reg_kernel", "fused local_level (3 products)", 3, f_sim, "mfma"),
        ("nr_mlp_kernel", "token scorers (4 token sets)", 4, f_mlp, "mfma"),
        ("nr_linear_group_kernel<2, 2, 1, 2, true, true>", "clustering stage 0: k=3 token convolution (split-bf16)", 1, 2 * r0 * 3 * d * d, "mfma"),
        ("nr_back_beside_linear_kernel", "clustering stage 0: kv projection (split-bf16) + DPC-KNN / merge workgroups", 1, 2 * r0 * d * 2 * d, "mfma"),
        ("nr_linear_group_kernel<1, 2, 4, 2, true, true>", "clustering stage 1: token convolution", 1, 2 * r1 * 3 * d * d, "mfma"),
        ("nr_linear_group_kernel<1, 2, 2, 2, false, true>", "clustering stage 1: q + kv projections", 1, 2 * (r1 * 2 + q1) * d * d, "mfma"),
        ("nr_linear_group_kernel<1, 2, 4, 2, false, true>", "clustering: q (stage 0) and the two proj GEMMs", 3, 2 * (2 * q0 + q1) * d * d, "mfma"),
        ("nr_group_front2_kernel", "clustering stage 0: LayerNorm, score, norm1, pairwise distances", 1, None, "vector / latency"),
        ("nr_group_front_back2_kernel", "clustering stage 1: the same + DPC-KNN + merge", 1, None, "latency"),
        ("nr_group_attention", "clustering: score-biased attention (both stages)", 2, None, "latency"),
        ("nr_prepare_pair_kernel", "normalise + bf16 split of the batch tokens", 1, None, "hbm"),
        ("nr_sinkhorn_small_kernel", "Sinkhorn 50 iterations + uniform CE rows (2 workgroups)", 1, None, "latency"),
        ("nr_row_losses_fwd_kernel", "top-K + centrality / neighbour / KL rows (32 workgroups)", 1, None, "latency"),
    ]


def kernel_profiles(args):
    """N = 1: two children under `rocprofv3 --kernel-trace --stats`, started BEFORE this process touches the GPU (a child is a
    fresh process; nothing is exec'ed over a GPU-initialised one): (a) this very command's step loop (`--_child`: warm-ups,
    clock ramp, the K timed steps and the repeats -- no parity step, no roofline launches, no CPU baseline), (b) the three
    similarity launches alone (tools/roofline_launches.py).  Returns {"in_step": {name: (calls, avg_us)}, "alone": {...}} and
    leaves the two kernel_stats.csv under gpurun_out/ (copied to profiles/ by hand when they are to be judged)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)

    def run(tag, argv):
        d = tempfile.mkdtemp(prefix="nr_prof_", dir="/tmp")
        cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "b", "--", sys.executable] + argv
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=900)
            files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
            if r.returncode != 0 or not files:
                print(f"[bench] kernel profile '{tag}' unavailable (rc {r.returncode}): {r.stderr[-400:]}", file=sys.stderr)
                return None
            keep = os.path.join(out_dir, f"bench_{tag}_kernel_stats.csv")
            shutil.copy(files[0], keep)
            rows = {}
            for row in csv.DictReader(open(keep)):
                rows[row["Name"]] = (int(row["Calls"]), float(row["AverageNs"]) / 1e3)
            return rows
        except (OSError, subprocess.SubprocessError) as e:
            print(f"[bench] kernel profile '{tag}' unavailable ({type(e).__name__}: {e})", file=sys.stderr)
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    me = os.path.abspath(__file__)
    child = [me, "--_child", "--no-cpu-baseline", "--no_kernel_profile", "--steps", str(args.steps), "--warmup", str(args.warmup),
             "--config", str(args.config), "--precision", args.precision, "--unroll", str(args.unroll)]
    child += ["--no-graph"] if args.no_graph else []
    child += ["--no-pipeline"] if args.no_pipeline else []
    res = {"in_step": run("step", child)}
    if args.config == 1:
        res["alone"] = run("alone", [os.path.join(ROOT, "tools", "roofline_launches.py")])
    return res


def launch_ranks(n):
    """`python bench.py --gpus N` typed without a launcher: start N fresh rank processes (one per GPU) through
    torch.distributed.run and relay their output.  This parent has made no GPU call (it only parsed arguments), and it
    does not exec: the ranks are children, their exit code becomes ours."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit(launch_ranks(args.gpus))
    prof = None
    if world == 1 and not args._child and not args.no_kernel_profile:
        prof = kernel_profiles(args)          # (children; this process has not touched the GPU yet)
    import torch.distributed as dist
    t_start = time.perf_counter()

    def progress(msg):
        """N > 1: a line per phase on stderr (rank 0 and the last rank), so that a run that stops shows where."""
        if world > 1 and rank in (0, world - 1):
            print(f"[bench] rank {rank} +{time.perf_counter() - t_start:6.1f} s: {msg}", file=sys.stderr, flush=True)
    watchdog = None
    if world > 1:
        # a multi-rank run that stops making progress (a collective one rank never joins) would otherwise sit there until the
        # caller's own limit: give up loudly after ten minutes instead
        import threading

        def _give_up():
            print(f"[bench] rank {rank}: no result after 600 s -- giving up (a collective that some rank never joined?)", file=sys.stderr, flush=True)
            os._exit(5)
        watchdog = threading.Timer(600.0, _give_up)
        watchdog.daemon = True
        watchdog.start()
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    n_joined = 1
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        # The job runs on a stream of the pool, not on the legacy default stream: replayed on the default stream, the exchange
        # graphs of the step-interleaved job and the owner's loss graph on its own stream take turns instead of overlapping
        # (measured, W = 8 emulated: 723 us per round against 516; tools/rank_local_times.py).  N = 1: no difference (A/B, 3385 vs 3380).
        job_stream = torch.cuda.Stream()
        torch.cuda.set_stream(job_stream)
        # how many ranks the collective backend really joins: an all-reduce of ones (reported as `rccl_ranks`)
        ones_ = torch.ones(1, device=dev)
        dist.all_reduce(ones_)
        torch.cuda.synchronize()
        n_joined = int(round(float(ones_.item())))

    from neighborretr_amd import hip, ops, synth
    c = CFG
    if c["B"] % world:
        raise SystemExit(f"global batch {c['B']} must divide over the ranks")
    b = c["B"] // world
    model = build_model(args.precision, dev)
    model.config.world_size, model.config.local_rank = world, rank
    # N > 1, three forms of the job.  DEFAULT: step-interleaved (model.interleave_steps): every rank gathers every step's batch
    # and pushes it into its bank replica, the loss of step k is evaluated on rank k mod W with the full single-rank kernels --
    # loss-only steps depend on each other through the bank alone.  --sync_step: the sharded synchronous step.
    # --replicated_loss: the reference's form (every rank evaluates every loss).
    interleaved = world > 1 and not args.sync_step and not args.replicated_loss
    sharded = world > 1 and args.sync_step and not args.replicated_loss
    if world > 1:
        model.shard_loss = sharded
        model.interleave_steps = interleaved
        # the owner's loss beside the following steps: built and timed next to the serial form, the faster one is kept (below)
        model.interleave_overlap = interleaved and not args.no_overlap
    n_round = world if interleaved else 1          # steps after which every rank has evaluated a loss
    ctr = [0]                                      # the job's step counter (host side; identical on every rank)
    budget = [0]                                   # steps the caller still wants: a replay may cover several (see --unroll)
    full = synth.make_problem(1002, c["B"], c["Nt"], c["Nv"], c["M"])
    sl = slice(rank * b, (rank + 1) * b)
    shard = {k: torch.from_numpy(full[k][sl]).to(dev) for k in ("text_feat", "video_feat", "text_mask", "video_mask", "idx")}
    bank = {k: torch.from_numpy(full[k]).to(dev) for k in ("mb_feat_t", "mb_feat_v", "mb_mask_t", "mb_mask_v")}
    model.mb_feat_t, model.mb_feat_v = bank["mb_feat_t"], bank["mb_feat_v"]
    model.mb_mask_t, model.mb_mask_v = bank["mb_mask_t"], bank["mb_mask_v"]
    model.mb_ind = torch.arange(10 ** 6, 10 ** 6 + c["M"], device=dev)
    result = {}          # the step's [5] loss vector; inside a captured graph it lives in the graph's own pool

    def _as_vector(losses):
        base = losses[0]._base                  # the five scalars are views of one [5] tensor
        return base if base is not None and base.numel() == 5 else torch.stack(losses)

    def step():
        """The WHOLE step through the model's forward: [N>1: packed all-gather ->] losses -> bank push.  (Interleaved: the
        losses on the step's owner, exchange + bank push on the other ranks.)"""
        if interleaved:
            model._step_index = ctr[0]
        ctr[0] += 1
        with torch.no_grad():
            losses = model(shard["text_feat"], shard["text_mask"], shard["video_feat"], shard["video_mask"], shard["idx"], 0)
            if losses is not None:
                result["losses"] = _as_vector(losses)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- HIP-graph capture.  N = 1: the whole step.  N > 1, sharded loss (default): in order of preference
    #   "whole"      the whole step INCLUDING its collectives (packed all-gather, the clustering's max exchange, the gathers of
    #                global tokens / centralities, the row-term all-reduce) as ONE graph -- RCCL collectives capture and replay on
    #                this runtime (tools/rccl_capture_probe.py); trusted only after a replay has reproduced the eager step;
    #   "segmented"  the rank-local segments between the collectives as graphs, the collectives eager between their replays
    #                (comm.SegmentedStep): 6 replays + 5 collectives instead of ~60 eager launches;
    #   "eager".
    # Every decision is COLLECTIVE: a rank records its own verdict and never leaves the common sequence of collectives on its
    # own -- the ranks agree (MIN over a gloo side group with a short timeout) after each capture and after each validation,
    # and all keep a form or all drop to the next one.  N > 1, --replicated_loss: the exchange step eager into static buffers,
    # the loss replayed from a graph (round 2's form).
    graph = None
    static = None
    unroll_check = None
    step_form, n_segments = "eager", 0
    whole_step_graph = world == 1
    if world > 1 and args.replicated_loss:
        from neighborretr_amd.dist import packed_allgather

        def gather():
            return packed_allgather(shard["text_feat"], shard["video_feat"], shard["idx"], shard["text_mask"],
                                    shard["video_mask"], model.config)
        # the unpack kernel of the exchange writes straight into these static buffers (what the graph reads)
        static = [t.clone() for t in gather()]
        model.config._gather_out = tuple(static)

        def after_gather():
            with torch.no_grad():
                tf, vf, ix, tm, vm = static
                result["losses"] = _as_vector(model.loss_step(tf, vf, tm, vm, ix))

        def step():                                        # noqa: F811
            with torch.no_grad():
                gather()
            (graph.replay if graph is not None else after_gather)()
    progress("process group up, model built; eager warm-up steps")
    for _ in range(3 * n_round):
        step()
    torch.cuda.synchronize()
    progress("eager steps done")
    # launches of one eager step: C-ABI entry-point calls (a grouped stage call = up to 7 kernels); interleaved: of one round
    # of W steps on this rank (one loss evaluation + W - 1 exchange-and-push steps), divided by W
    before = hip.N_CALLS
    for _ in range(n_round):
        step()
    torch.cuda.synchronize()
    abi_calls = round((hip.N_CALLS - before) / n_round, 1) if rank == 0 else None
    run = step
    if world > 1 and not args.replicated_loss and not args.no_graph:
        from neighborretr_amd import comm
        cc = comm.CollectiveCapture(world, rank, log=lambda msg: print(f"[bench] {msg}", file=sys.stderr))
        rng = model._rng_state_on(dev)

        seen = {"ok": True}

        def eager_pass():
            ctr[0] = 0
            rng[1] = 4242                    # the DPC-KNN tie-break noise is a function of this counter: the same draws for both passes
            for _ in range(n_round):
                step()
            torch.cuda.synchronize()
            seen["eager"], seen["ok"] = result["losses"].clone(), True

        def attempt(make, what):
            """make() -> (replay, keep-alive): a replayable form of `step`, validated against the eager step on every rank
            (comm.CollectiveCapture.attempt) -> the form or None, on every rank alike."""
            def make_pass():
                replay, keep = make()

                def replay_pass():
                    # two rounds: the first step by step (the owner / other graphs), the second as ONE replay (the round graph);
                    # with the bank frozen and the noise counter rewound both must reproduce the eager round
                    for allowed in ((0, n_round) if (interleaved and args.round_graph) else (0,)):
                        ctr[0] = 0
                        rng[1] = 4242
                        budget[0] = allowed
                        done = 0
                        while done < n_round:
                            done += replay() or 1
                        torch.cuda.synchronize()       # (each pass is held against the eager round; the verdict is agreed on by all ranks)
                        seen["ok"] = seen["ok"] and bool(torch.allclose(result["losses"], seen["eager"], rtol=1e-5, atol=1e-6))
                return replay_pass, (replay, keep)
            form = cc.attempt(what, eager_pass, make_pass, lambda: result["losses"],
                              lambda a, b_: seen["ok"] and torch.allclose(a, b_, rtol=1e-5, atol=1e-6),
                              freeze=lambda on: setattr(model, "bank_frozen", on))
            return None if form is None else form[1]

        def per_phase(capture_one):
            """The replayable form of `step`: one capture, or -- interleaved -- two: the step this rank owns and the step it
            does not; replay() picks by the job's step counter, as the eager step does."""
            if not interleaved:
                g = capture_one(step)
                return g.replay, g
            forms, outs = {}, {}
            for own in (True, False):
                ctr[0] = rank if own else rank + 1
                if own and model.interleave_overlap:
                    # the owned step in two halves (neighborretr_amd.interleave): exchange + bank copy + push on this stream,
                    # the loss from the copy on the model's loss stream, beside the following steps
                    from neighborretr_amd.interleave import OverlappedOwnedStep

                    def exchange_half(slot_index):
                        model._step_index = rank
                        return model.owned_exchange(shard["text_feat"], shard["text_mask"], shard["video_feat"], shard["video_mask"],
                                                    shard["idx"], slot_index=slot_index)
                    forms[own] = OverlappedOwnedStep(model, exchange_half, capture_one)
                    outs[own] = forms[own].losses
                    continue
                forms[own] = capture_one(step)
                outs[own] = result.get("losses")
            if args.round_graph:                 # one whole round (W consecutive steps: this rank's own and the W - 1 others)
                ctr[0] = 0

                def one_round():
                    for _ in range(world):
                        step()
                forms["round"] = capture_one(one_round)
                outs["round"] = result.get("losses")

            def replay():
                """The next step; a whole round at once when the counter stands at a round's start and the caller allows it."""
                if "round" in forms and ctr[0] % world == 0 and budget[0] >= world:
                    ctr[0] += world
                    budget[0] -= world - 1
                    forms["round"].replay()
                    result["losses"] = outs["round"]
                    return world
                own = ctr[0] % world == rank
                ctr[0] += 1
                forms[own].replay()
                if own:
                    # (each graph writes the losses into its own pool; the overlapped form has one loss graph per slot)
                    result["losses"] = getattr(forms[True], "losses", outs[True])
                return 1
            return replay, forms

        def make_whole():
            if args.fail_whole_capture:          # rehearsal: rank 0 "fails", the others "succeed" -- all must end up segmented
                if rank == 0:
                    raise RuntimeError("--fail_whole_capture")
                return (lambda: None), None

            def one(fn):
                # with collectives inside the capture other threads of the process (the process group's watchdog) may touch
                # the runtime while this thread captures: thread-local capture mode keeps their calls out of its error checking
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    fn()
                return g
            return per_phase(one)

        def make_segmented():
            return per_phase(lambda fn: comm.SegmentedStep(fn).capture())

        def segments_of(keep):
            return keep[True].n_segments if interleaved else keep.n_segments
        # Interleaved: every kind of form (whole / segmented) is built with the owner's loss BESIDE the following steps
        # (model.interleave_overlap) and serial, each validated against the eager steps, and the FASTEST of them is kept -- decided
        # together: a few rounds of each are timed and the ranks agree on the maximum over ranks (the job runs at its slowest rank's
        # pace).  How well the loss graphs and the exchange graphs overlap depends on which hardware queues their streams land on
        # (measured on one GPU, W = 8 emulated: 362-470 us per round against 500 serial, and 2 ms with an unlucky draw under other
        # queue counts), so the overlapped form is built up to --overlap_tries times on fresh streams.
        overlap_on = bool(model.interleave_overlap)
        probes = []

        def probe(form_):
            """us per step of a validated form over 6 rounds (after 2 warm ones), maximum over the ranks."""
            import gc
            replay_ = form_[0]
            gc.collect()                     # (discarded forms own HIP graphs: destroying them mid-probe costs up to a second)
            gc_on = gc.isenabled()
            gc.disable()
            try:
                ctr[0] = 0
                for _ in range(2 * n_round):
                    replay_()
                sync()
                t0_ = time.perf_counter()
                for _ in range(6 * n_round):
                    replay_()
                sync()
            finally:
                if gc_on:
                    gc.enable()
            t_ = torch.tensor([(time.perf_counter() - t0_) / (6 * n_round) * 1e6], device=dev, dtype=torch.float64)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            return float(t_.item())

        def candidates(make, what):
            """[(us per step, overlap?, form)] of the forms `make` yields: overlapped (several draws of streams) and serial."""
            out_ = []
            for ov in ((True, False) if overlap_on else (False,)):
                model.interleave_overlap = ov
                for _try in range(max(1, args.overlap_tries) if ov else 1):
                    if ov:
                        model._owned_ring, model._owned = [], None         # fresh slots: fresh loss streams
                        model.owned_slots = 1 if _try == 3 else 2          # (the fourth draw: one slot -- sometimes the better one at W = 8)
                        for _ in range(n_round):
                            step()
                        torch.cuda.synchronize()
                    progress(f"building the {what} form" + (f", owner's loss beside the following steps (draw {_try + 1})" if ov else ""))
                    f_ = attempt(make, what + (" (loss beside the following steps)" if ov else ""))
                    progress("  ... " + ("validated" if f_ is not None else "not available"))
                    if f_ is None:
                        break
                    out_.append([None, ov, f_])
            if len(out_) > 1:
                for c_ in out_:
                    c_[0] = probe(c_[2])
                    probes.append({"owner_loss_beside": c_[1], "slots": len(getattr(c_[2][1][True], "pairs", ())) if c_[1] else None, "us_per_step": round(c_[0], 1)})
                out_.sort(key=lambda c_: c_[0])
                progress("probed: " + ", ".join(f"{c_[0]:.0f} us/step" + (" (beside)" if c_[1] else " (serial)") for c_ in out_))
                # the form that is timed is the MEDIAN of the validated draws (upper median of an even count), not the fastest: the
                # headline is then not a maximum over draws of the quantity it reports; every draw is listed in `form_probes`
                mid = out_[len(out_) // 2]
                out_ = [mid] + [c_ for c_ in out_ if c_ is not mid]
                progress(f"timing the median draw: {mid[0]:.0f} us/step" + (" (beside)" if mid[1] else " (serial)"))
            return out_
        form = None
        if args.backend == "nccl" or args.fail_whole_capture:
            cands = candidates(make_whole, "whole-step")
            if cands:
                _, ov, form = cands[0]
                model.interleave_overlap = ov
        if form is not None:
            step_form, graph = "whole", form[1]
        else:
            # a side stream cannot stay forked across a cut between two segments: the synchronous sharded step runs on one
            # stream in this form; the interleaved step's only collective comes before anything is forked
            model.use_side_streams = interleaved
            for _ in range(2 * n_round):
                step()
            torch.cuda.synchronize()
            cands = candidates(make_segmented, "segmented")
            if cands:
                _, ov, form = cands[0]
                model.interleave_overlap = ov
                step_form, graph, n_segments = "segmented", form[1], segments_of(form[1])
            else:
                model.use_side_streams = True
                model.interleave_overlap = False
        if form is not None:
            run = form[0]
        ctr[0] = 0
    elif not args.no_graph:
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                (step if whole_step_graph else after_gather)()
            graph = g
            step_form = "whole" if whole_step_graph else "exchange eager + loss graph"
            if whole_step_graph:
                run = g.replay
                if args.unroll > 1:              # U consecutive steps as one graph (their dependencies kept), K mod U singly
                    out1 = result["losses"]
                    outs_u = []
                    pipes_u = []

                    def unrolled(pipelined):
                        """U steps into the running capture.  pipelined: a step's tail (Sinkhorn solve, row losses, bank push) stays
                        forked until the capture ends; the next step starts behind this step's logits and bank push
                        (modeling.StepPipeline)."""
                        from neighborretr_amd.capture_guard import record_event, wait_event, wait_stream
                        from neighborretr_amd.modeling import StepPipeline
                        del outs_u[:]
                        if not pipelined:
                            for _ in range(args.unroll):
                                step()
                                outs_u.append(result["losses"])
                            return
                        origin, prev, pending = torch.cuda.current_stream(), None, []
                        del pipes_u[:]
                        try:
                            for k in range(args.unroll):
                                early = record_event(origin)                 # where this step's local branch forks from: in front of ...
                                if prev is not None and prev.push_done is not None and not args.decouple_push:
                                    wait_event(origin, prev.push_done)       # ring head and bank rows: the one dependency between two steps
                                model._pipeline = prev = StepPipeline(k, prev, decoupled=args.decouple_push)
                                prev.early_fork = early
                                pipes_u.append(prev)                         # (owns its step's buffers until the capture has ended)
                                step()                                       # prologue -> clustering -> logits on the origin; the rest forked
                                outs_u.append(result["losses"])
                                pending += prev.pending
                        finally:
                            model._pipeline = None
                        for st_ in pending:
                            wait_stream(origin, st_)

                    def capture_unrolled(pipelined):
                        model.prepare_pipeline(args.unroll, c["B"])
                        gu = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gu):
                            unrolled(pipelined)
                        del pipes_u[:]                 # the capture has ended: the steps' buffers go back to the graph's own pool
                        return gu, list(outs_u)

                    def state():
                        sh = model._mb_shadow or ()
                        return [t_ for t_ in ([model._mb[k_] for k_ in model._mb] + [t_ for p_ in sh for t_ in (p_.hi, p_.lo, p_.norm)]
                                              + [model._mb_head_dev, model._rng_state]) if t_ is not None]

                    def equals_single_steps(gu, outs):
                        """The unrolled graph against U replays of the single-step graph from the same bank / ring / noise state:
                        every step's losses and the state left behind, bit for bit."""
                        saved = [t_.clone() for t_ in state()]
                        gu.replay()
                        torch.cuda.synchronize()
                        got, after = [o.clone() for o in outs], [t_.clone() for t_ in state()]
                        for t_, s_ in zip(state(), saved):
                            t_.copy_(s_)
                        want = []
                        for _ in range(args.unroll):
                            g.replay()
                            torch.cuda.synchronize()
                            want.append(out1.clone())
                        same = all(torch.equal(a_, b_) for a_, b_ in zip(got, want)) and all(torch.equal(a_, b_) for a_, b_ in zip(after, state()))
                        return same, max(float((a_ - b_).abs().max()) for a_, b_ in zip(got, want))
                    gU, outs = capture_unrolled(not args.no_pipeline)
                    unroll_form = "sequential" if args.no_pipeline else "pipelined"
                    ok_, dl_ = equals_single_steps(gU, outs)
                    if not ok_ and not args.no_pipeline:
                        print(f"[bench] the pipelined {args.unroll}-step graph differs from single-step replays (max |dL| {dl_:.2e}): taking the "
                              "sequential form", file=sys.stderr)
                        gU, outs = capture_unrolled(False)
                        unroll_form = "sequential (the pipelined form failed its check)"
                        ok_, dl_ = equals_single_steps(gU, outs)
                    unroll_check = {"steps_per_graph": args.unroll, "form": unroll_form, "equals_single_step_replays": bool(ok_), "max_dL": dl_}
                    outU = outs[-1]
                    graph = (g, gU)
                    use_unrolled = bool(ok_)             # (a graph that does not reproduce the single steps is never timed)

                    def run():                   # noqa: F811
                        if use_unrolled and budget[0] >= args.unroll:
                            budget[0] -= args.unroll - 1
                            gU.replay()
                            result["losses"] = outU      # (each graph writes the losses into its own pool)
                            return args.unroll
                        g.replay()
                        result["losses"] = out1
                        return 1
        except Exception as e:          # graphs are an optimisation, never a requirement
            import traceback
            traceback.print_exc()       # (with the chain: an exception inside a capture is followed by capture_end's own)
            print(f"[bench] graph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graph, run = None, step
            torch.cuda.synchronize()

    # Clock ramp: the first ~0.1 s of back-to-back steps after an idle period run 10 % slower than the steady state (measured:
    # 0.361 ms/step for the first 200 steps after a 20-step warm-up, 0.324 for every later 200) -- the chip has to leave its
    # idle power state.  Part of the untimed set-up, like the capture warm-ups above: >= 0.4 s of steps before the contract's
    # W warm-up steps, so that the timed K steps measure the steady state whatever W is.
    def run_steps(n):
        """EXACTLY n steps: replays that cover several steps (--unroll) only while that many are still wanted."""
        budget[0] = n
        while budget[0] > 0:
            run()
            budget[0] -= 1
    # Everything has been built: drop what is unreachable NOW (the discarded forms of the attempts above own HIP graphs and
    # streams; destroying them takes the runtime up to a second) and keep the collector out of the timed steps
    import gc
    gc.collect()
    torch.cuda.synchronize()
    gc.disable()
    t_ramp = time.perf_counter()
    if world == 1:
        while time.perf_counter() - t_ramp < 0.4:
            run_steps(48)
            torch.cuda.synchronize()
    else:                                   # the SAME number of steps on every rank: a clock-bounded loop would let the ranks disagree
        run_steps(1200)
        torch.cuda.synchronize()
    progress(f"form: {step_form}; clock ramp done; timing {args.steps} steps")
    run_steps(args.warmup)
    if interleaved:                         # the timed steps start at a round's start on every rank
        run_steps((-ctr[0]) % world)
    sync()
    t0 = time.perf_counter()
    run_steps(args.steps)
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    losses = result["losses"].cpu().numpy().tolist()
    # Spread of the timed loop (extra fields; `value` / `steps` / `ms_per_step` above keep the contract's meaning): the same
    # K-step loop repeated, each repeat bracketed like the timed region.  At the driver's --steps 20 the region is ~7 ms, so
    # one number says little about its own noise.
    rep_ms = []
    for _ in range(max(5, min(25, int(200 / max(args.steps, 1))))):
        sync()
        t1 = time.perf_counter()
        run_steps(args.steps)
        sync()
        tr = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        rep_ms.append(float(tr.item()) / args.steps * 1e3)

    if args._child:                          # a profiled child (kernel_profiles): the step loop is all it is for
        print(json.dumps({"child": True, "steps": args.steps, "ms_per_step": round(dt / args.steps * 1e3, 4)}))
        return
    extra = {}
    out_sync = {}

    def sync_probe():
        nonlocal interleaved, sharded, n_round, overlap_on, run, step_form, n_segments, graph
        # ---- second timed field: the SYNCHRONOUS sharded step (row slabs of S, 1/W of the bank and clustering work,
        # five collectives per step) -- the form a training step can use -- in the same job, same ranks, same K steps.  The
        # closures above read `interleaved` / `sharded` / `n_round` when they are called: flipped here, they build that form.
        progress("second field: the synchronous sharded step")
        primary = dict(step_form=step_form, n_segments=n_segments, overlap=model.interleave_overlap, run=run, graph=graph)
        interleaved, sharded, n_round = False, True, 1
        model.interleave_steps, model.interleave_overlap, model.shard_loss, model.use_side_streams = False, False, True, True
        overlap_on = False
        run2, form2_name, nseg2 = step, "eager", 0
        try:
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            form2 = None
            if args.backend == "nccl" or args.fail_whole_capture:
                cands2 = candidates(make_whole, "whole-step [sync]")
                if cands2:
                    form2, form2_name = cands2[0][2], "whole"
            if form2 is None:
                model.use_side_streams = False
                for _ in range(2):
                    step()
                torch.cuda.synchronize()
                cands2 = candidates(make_segmented, "segmented [sync]")
                if cands2:
                    form2, form2_name = cands2[0][2], "segmented"
                    nseg2 = segments_of(form2[1])
                else:
                    model.use_side_streams = True
            if form2 is not None:
                run2 = form2[0]
            run = run2
            ctr[0] = 0
            run_steps(max(args.warmup, 3))
            sync()
            t1 = time.perf_counter()
            run_steps(args.steps)
            sync()
            ts_ = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
            dist.all_reduce(ts_, op=dist.ReduceOp.MAX)
            dts = float(ts_.item())
            out_sync["sync_step"] = {"what": "the synchronous sharded step (every rank takes part in every loss: row slabs of S, 1/W of the bank and "
                                          "clustering work, five collectives per step) timed in the same job over the same K steps",
                                  "value": round(args.steps / dts, 2), "unit": "steps/s", "ms_per_step": round(dts / args.steps * 1e3, 4),
                                  "step_form": form2_name, "graph_segments": nseg2,
                                  "losses": [round(float(x), 5) for x in result["losses"].cpu().numpy().tolist()]}
        except Exception as e:               # the headline never depends on the second field
            import traceback
            traceback.print_exc()
            out_sync["sync_step"] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.synchronize()
        finally:
            # back to the primary form for what follows (exchange timing, parity gate)
            interleaved, sharded, n_round = True, False, world
            model.interleave_steps, model.shard_loss = True, False
            model.interleave_overlap = primary["overlap"]
            model.use_side_streams = True
            step_form, n_segments, run, graph = primary["step_form"], primary["n_segments"], primary["run"], primary["graph"]

    if world > 1:
        # the exchange step alone (pack, packed all-gather, unpack; eager), so that a rehearsal on gloo ranks -- whose collectives
        # go through the host -- shows how much of a step is the collective's own data path
        from neighborretr_amd.dist import packed_allgather as _pg
        with torch.no_grad():
            for _ in range(3):
                _pg(shard["text_feat"], shard["video_feat"], shard["idx"], shard["text_mask"], shard["video_mask"], model.config)
            sync()
            t1 = time.perf_counter()
            for _ in range(10):
                _pg(shard["text_feat"], shard["video_feat"], shard["idx"], shard["text_mask"], shard["video_mask"], model.config)
            sync()
        extra["exchange_step_ms"] = round((time.perf_counter() - t1) / 10 * 1e3, 4)
    if args.backward and world == 1:
        tf = shard["text_feat"].clone().requires_grad_(True)
        vf = shard["video_feat"].clone().requires_grad_(True)

        def fb():
            model.zero_grad(set_to_none=True)
            tf.grad = vf.grad = None
            ls = model(tf, shard["text_mask"], vf, shard["video_mask"], shard["idx"], 0)
            ls[0].backward()
        for _ in range(5):
            fb()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nfb = max(10, args.steps // 10)
        for _ in range(nfb):
            fb()
        torch.cuda.synchronize()
        extra["fwd_bwd_ms_per_step"] = (time.perf_counter() - t1) / nfb * 1e3
        # the same training step (forward + backward, grads left in .grad) captured as ONE HIP graph: the eager
        # step is bound by the host issuing the autograd-traced clustering ops, the replay is not
        if not args.no_graph:
            try:
                with model.graph_capture_mode():
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        for _ in range(3):
                            fb()
                    torch.cuda.current_stream().wait_stream(side)
                    torch.cuda.synchronize()
                    gfb = torch.cuda.CUDAGraph()
                    model.zero_grad(set_to_none=True)
                    tf.grad = vf.grad = None
                    with torch.cuda.graph(gfb):
                        ls = model(tf, shard["text_mask"], vf, shard["video_mask"], shard["idx"], 0)
                        ls[0].backward()
                for _ in range(5):
                    gfb.replay()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(nfb):
                    gfb.replay()
                torch.cuda.synchronize()
                extra["fwd_bwd_graph_ms_per_step"] = (time.perf_counter() - t1) / nfb * 1e3
            except Exception as e:
                print(f"[bench] fwd+bwd graph capture unavailable ({type(e).__name__}: {e})", file=sys.stderr)
                torch.cuda.synchronize()

    if args.e2e and world == 1:
        extra["e2e"] = e2e_bench(dev)

    if args.emulate_world is not None and world == 1:
        # what one rank of the sharded step would do at W ranks, on this GPU (extra field; tools/rank_local_times.py)
        from tools import rank_local_times as RL
        RL.init_one_rank_group()
        keep = (model.config.world_size, model.config.local_rank, model.shard_loss)
        lines_ = []
        full_dev = {k: torch.from_numpy(v).to(dev) for k, v in full.items()}
        try:
            extra["rank_local"] = {"what": "rank 0 of the sharded loss-only step at W ranks, emulated on one GPU: its own messages through a "
                                           "1-rank RCCL communicator, the peers' parts pre-filled; no wire time (tools/rank_local_times.py)",
                                   "per_world": [RL.measure(model, full_dev, W, 0, dev, lines_) for W in (args.emulate_world or [2, 4, 8])]}
        finally:
            model.config.world_size, model.config.local_rank, model.shard_loss = keep
        dist.destroy_process_group()

    # ---- roofline of the dominant kernel, measured live with HIP events on the launch stream ------------
    roofline = None
    if rank == 0:
        f_sim, f_mlp = algorithmic_flops(c["B"], c["Nt"], c["Nv"], c["M"])
        from neighborretr_amd import head
        p_bb, p_mlp, p_bank = head.precision_plan(model._prec())
        B, Nt, Nv, M = c["B"], c["Nt"], c["Nv"], c["M"]
        allf = {k: torch.from_numpy(full[k]).to(dev) for k in ("text_feat", "video_feat", "text_mask", "video_mask")}
        with torch.no_grad():
            pt = ops.prepare_tokens(allf["text_feat"], allf["text_mask"])
            pv = ops.prepare_tokens(allf["video_feat"], allf["video_mask"])
            pbt = ops.prepare_tokens(bank["mb_feat_t"], bank["mb_mask_t"])
            pbv = ops.prepare_tokens(bank["mb_feat_v"], bank["mb_mask_v"])
            w = lambda n, N: torch.full((n, N), 1.0 / N, device=dev)
            w_t, w_v, w_bt, w_bv = w(B, Nt), w(B, Nv), w(M, Nt), w(M, Nv)

            # the step's similarity launches, as head.head_forward issues them: the batch x batch product, and the two bank
            # products either as one launch of chained tile pairs (nr_sim_pair_kernel) or as two
            paired = (head.PAIR_BANK_PRODUCTS and hip.local_level_group_kind(B, Nt, M, Nv, c["d"], p_bank) == 0
                      and hip.local_level_group_kind(M, Nt, B, Nv, c["d"], p_bank) == 0)
            n_launch = 2 if paired else 3

            def three():
                ops.local_level(pt, pv, w_t, w_v, B, Nt, B, Nv, p_bb, hip.OUT_FULL)
                if paired:
                    ops.local_level_group([(pt, pbv, w_t, w_bv, B, Nt, M, Nv, p_bank, hip.OUT_ROWSUM),
                                           (pbt, pv, w_bt, w_v, M, Nt, B, Nv, p_bank, hip.OUT_COLSUM)])
                else:
                    ops.local_level(pt, pbv, w_t, w_bv, B, Nt, M, Nv, p_bank, hip.OUT_ROWSUM)
                    ops.local_level(pbt, pv, w_bt, w_v, M, Nt, B, Nv, p_bank, hip.OUT_COLSUM)
            for _ in range(5):
                three()
            torch.cuda.synchronize()
            # replayed from a HIP graph so that the launches are back to back on the device (an eager loop
            # is host-bound once a launch is shorter than the Python call that issues it)
            replay, inner = three, 1
            if not args.no_graph:
                try:
                    g3 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g3):
                        for _ in range(10):          # a replay costs ~8 us of its own: amortised over 30 launches
                            three()
                    replay, inner = g3.replay, 10
                except Exception:
                    torch.cuda.synchronize()
            for _ in range(3):
                replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 50
            e0.record()
            for _ in range(reps):
                replay()
            e1.record()
            torch.cuda.synchronize()
        per_launch_s = e0.elapsed_time(e1) * 1e-3 / (n_launch * reps * inner)
        achieved = (f_sim / n_launch) / per_launch_s / 1e12
        traffic = mfma_busy = None
        pmc = next((q_ for q_ in (os.path.join(ROOT, "profiles", f"r0{k_}_pmc_sim.json") for k_ in (5, 4, 3, 2)) if os.path.exists(q_)),
                   os.path.join(ROOT, "profiles", "r03_pmc_sim.json"))
        pmc_name = "profiles/" + os.path.basename(pmc)
        if os.path.exists(pmc) and args.config == 1:     # PMC counters need rocprofv3 (separate passes): the committed passes are quoted here
            pj = json.load(open(pmc))
            traffic, mfma_busy = pj.get("bytes_per_launch_avg_over_step"), pj.get("mfma_busy_frac_flops_weighted")
        roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                    "traffic_source": pmc_name + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 x2 "
                                      "correction on FETCH_SIZE; bytes per launch, mean over the step's similarity launches; not re-measured in this run)",
                    "mfma_busy_frac": mfma_busy,
                    "mfma_busy_source": pmc_name + ": SQ_VALU_MFMA_BUSY_CYCLES / SIMDs / (SQ_BUSY_CYCLES / shader engines), "
                                        "weighted by the MFMA flops each launch issues",
                    "kernel": "nr_sim_reg_kernel (fused local_level; block shape chosen per launch for this shape), 3 launches/step" if args.config != 1 else
                              ("nr_sim_pair_kernel (fused local_level: the 2 bank products as chained 192x384 tile pairs through one "
                               "ping-pong K loop) + nr_sim_reg_kernel (the split-bf16 batch product on 96x192 blocks), both on 2x4 "
                               "waves = 2 launches/step") if paired else
                              ("nr_sim_reg_kernel (fused local_level: 2 bank products on 192x384 blocks, ping-pong K loop + the "
                               "split-bf16 batch product on 96x192 blocks, both on 2x4 waves = 3 launches/step)"),
                    "launches_per_step": n_launch,
                    "avg_launch_us": round(per_launch_s * 1e6, 2),
                    "algorithmic_flops_per_launch": f_sim / n_launch,
                    # the WHOLE step against the same peak: F_step = F_sim + F_mlp / ms_per_step.  The step
                    # is a latency chain (clustering -> logits -> Sinkhorn), not an MFMA-bound program: this is the number the
                    # steps/s target depends on, `frac` above is the similarity kernel alone.
                    "step_frac": round((f_sim + f_mlp) / (dt / args.steps) / 1e12 / PEAK_BF16_TFLOPS / world, 4),
                    "step_algorithmic_flops": f_sim + f_mlp}
        roofline["frac_hip_events_graph_of_30"] = roofline["frac"]
        roofline["timing"] = ("frac / achieved / avg_launch_us: HIP events on the launch stream around 50 replays of a graph of 10 x the "
                              "step's similarity launches, back to back, nothing else on the chip")
        if prof and prof.get("in_step"):
            # The same numbers from rocprofv3's kernel trace of THIS command's step loop (a child process under the profiler:
            # gpurun_out/bench_step_kernel_stats.csv) and of the similarity launches alone (bench_alone_kernel_stats.csv):
            # `frac_in_step` is what the kernels reach INSIDE the timed region, beside the other branch's kernels; `frac_alone`
            # is the per-kernel average of the profiler for the launches alone.  `frac` above follows from neither file.
            def pick(rows, key):
                hit = [(n_, v_) for n_, v_ in rows.items() if key in n_]
                calls = sum(v_[0] for _, v_ in hit)
                return calls, (sum(v_[0] * v_[1] for _, v_ in hit) / calls if calls else None)
            ins = prof["in_step"]
            # (whether the batch's two scorers really ran as ONE paired launch at this config -- read off the trace)
            roofline["paired_batch_scorer_launch_taken"] = any("nr_mlp_pair_kernel" in n_ for n_ in ins)
            sim_calls, sim_us = pick(ins, "nr_sim_reg_kernel")
            pair_calls, pair_us = pick(ins, "nr_sim_pair_kernel")
            n_steps_traced = (sim_calls + pair_calls) / n_launch if (sim_calls + pair_calls) else 0
            if n_steps_traced:
                t_sim = (sim_calls * (sim_us or 0.0) + pair_calls * (pair_us or 0.0)) / n_steps_traced       # us of similarity kernels per step
                roofline["frac_in_step"] = round(f_sim / (t_sim * 1e-6) / 1e12 / PEAK_BF16_TFLOPS, 4)
                roofline["avg_launch_us_in_step"] = round(t_sim / n_launch, 2)
                roofline["in_step_source"] = ("gpurun_out/bench_step_kernel_stats.csv: rocprofv3 --kernel-trace --stats of this command's "
                                              f"step loop ({int(n_steps_traced)} steps traced)")
                if args.config == 1:
                    ks = []
                    for key, role, per_step, flops, bound in kernel_table(c):
                        calls, us = pick(ins, key)
                        if not calls:
                            continue
                        per = calls / n_steps_traced
                        e_ = {"kernel": key, "role": role, "launches_per_step": round(per, 2), "avg_us": round(us, 2),
                              "us_per_step": round(us * per, 1), "bound": bound}
                        if flops:
                            tf_ = flops / (us * per * 1e-6) / 1e12
                            e_.update(algorithmic_flops_per_step=flops, tflops=round(tf_, 1), frac_of_bf16_peak=round(tf_ / PEAK_BF16_TFLOPS, 4))
                        ks.append(e_)
                    ks.sort(key=lambda e_: -e_["us_per_step"])
                    roofline["kernels"] = ks
                    roofline["kernels_note"] = ("every kernel of the step with >= 1 % of its kernel time, in-step averages from the same trace; "
                                                "algorithmic flops per SURVEY 8d (split-bf16 passes are not counted extra); us_per_step adds "
                                                "up to more than ms_per_step: the step's two branches run beside each other")
            if prof.get("alone"):
                a_calls, a_us = pick(prof["alone"], "nr_sim_reg_kernel")
                if a_calls:
                    roofline["frac_alone"] = round((f_sim / n_launch) / (a_us * 1e-6) / 1e12 / PEAK_BF16_TFLOPS, 4)
                    roofline["avg_launch_us_alone"] = round(a_us, 2)
                    roofline["alone_source"] = ("gpurun_out/bench_alone_kernel_stats.csv: rocprofv3 --kernel-trace --stats of "
                                                "tools/roofline_launches.py (the three products alone, 1650 launches)")
                    # `frac` / `achieved` / `avg_launch_us` = the profiler's per-kernel average of the launches alone: the figure a
                    # reader recomputes from the committed kernel_stats.csv (profiles/); the HIP-event figure stays beside it
                    roofline["hip_events_avg_launch_us"] = roofline["avg_launch_us"]
                    roofline["frac"], roofline["avg_launch_us"] = roofline["frac_alone"], roofline["avg_launch_us_alone"]
                    roofline["achieved"] = round(roofline["frac_alone"] * PEAK_BF16_TFLOPS, 2)
                    roofline["timing"] = ("frac / achieved / avg_launch_us: rocprofv3 per-kernel average of the step's three similarity launches "
                                          "ALONE (alone_source); frac_in_step: the same kernels inside the timed step loop (in_step_source); "
                                          "frac_hip_events_graph_of_30: HIP events around 50 replays of a graph of 10 x the three launches")
                    # The three products one by one.  `frac` above averages them at their ALGORITHMIC flops; the batch x batch product of
                    # the mixed plan runs split-bf16 (three MFMA passes for fp32-grade logits: a third of what it issues counts), the two
                    # bank products one bf16 pass.  The kernel's template arguments tell them apart in the trace (5th: split-bf16 tile);
                    # only reported when the trace shows exactly the expected split of launches.
                    B_, Nt_, Nv_, M_, d_ = c["B"], c["Nt"], c["Nv"], c["M"], 512
                    prods = [("batch text x batch video", 2.0 * d_ * (B_ * Nt_) * (B_ * Nv_), args.precision != "bf16_all"),
                             ("batch text x bank video", 2.0 * d_ * (B_ * Nt_) * (M_ * Nv_), args.precision == "bf16x3"),
                             ("bank text x batch video", 2.0 * d_ * (M_ * Nt_) * (B_ * Nv_), args.precision == "bf16x3")]

                    def is_split(name):
                        targs = name.split("<", 1)[1].split(">", 1)[0].split(",") if "<" in name else []
                        return len(targs) > 4 and targs[4].strip() == "true"
                    rows_ = [(n_, v_) for n_, v_ in prof["alone"].items() if "nr_sim_reg_kernel" in n_]
                    by = []
                    for split in (True, False):
                        mine = [p_ for p_ in prods if p_[2] == split]
                        hit = [(n_, v_) for n_, v_ in rows_ if is_split(n_) == split]
                        calls = sum(v_[0] for _, v_ in hit)
                        if not mine:
                            continue
                        if not calls or abs(calls / a_calls - len(mine) / 3.0) > 0.02:
                            by = None                        # (another tile form took this product -- e.g. three passes on the one-pass tile)
                            break
                        us = sum(v_[0] * v_[1] for _, v_ in hit) / calls
                        fl = sum(p_[1] for p_ in mine) / len(mine)
                        by.append({"products": [p_[0] for p_ in mine], "arithmetic": "split-bf16 (3 MFMA passes)" if split else "one bf16 pass",
                                   "launches_per_step": len(mine), "avg_launch_us": round(us, 2), "algorithmic_flops_per_launch": fl,
                                   "frac": round(fl / (us * 1e-6) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                   "frac_of_issued_mfma_flops": round((3 if split else 1) * fl / (us * 1e-6) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                   "kernel_variants": [n_.split("(")[0].replace("void ", "") for n_, _ in hit]})
                    if by:
                        roofline["by_product"] = by

    if rank == 0:
        line = {
            "metric": f"sim+loss steps/sec at global B={c['B']}, d=512",
            "value": round(args.steps / dt, 2), "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "abi_calls_per_rank_step": abi_calls,
            "rccl_ranks": n_joined if world > 1 else None,
            "collective_backend": (args.backend + (" (= RCCL)" if args.backend == "nccl" else "")) if world > 1 else None,
            "config": {"workload": c["name"] + ", loss-only forward incl. token clustering and bank push",
                       "global_batch": c["B"], "per_rank_batch": b, "precision_plan": args.precision,
                       "hip_graph": graph is not None, "step_form": step_form, "graph_segments": n_segments,
                       "unrolled_graph": unroll_check,
                       "steps_per_graph": (1 if (graph is None or args.unroll == 1 or step_form == "exchange eager + loss graph")
                                           else ((world if args.round_graph else 1) if interleaved else (args.unroll if world == 1 else 1))),
                       "parallelism": f"dp{world}: " + (
                           ("packed all-gather every step on every rank, bank replicated by pushing every gathered batch; the loss of step k "
                            "evaluated on rank k mod W with the single-rank kernels (step-interleaved: loss-only steps depend on each other "
                            "through the bank alone)" if interleaved else
                            "packed all-gather + loss, bank and clustering work sharded over the ranks, five collectives per step (synchronous)"
                            if sharded else "packed all-gather + replicated loss; exchange step eager, loss from a HIP graph")
                           + "; " + {"whole": "collectives inside the HIP graph(s)", "segmented": f"{n_segments} rank-local segments per step as HIP "
                                     "graphs, the collectives eager between them", "eager": "launched eagerly",
                                     "exchange eager + loss graph": "exchange eager + loss graph"}[step_form]) if world > 1 else "dp1",
                       "owner_loss": (("beside the following steps, from a copy of the bank (two graphs per owned step)" if model.interleave_overlap
                                       else "in front of the following steps") if interleaved else None),
                       "form_probes": (probes or None) if world > 1 and not args.replicated_loss and not args.no_graph else None,
                       "memory_bank": "ring (device head) + persistent prepared bf16 shadow, extended by the batch rows at every push"},
            "losses": [round(float(x), 5) for x in losses],
            "ms_per_step_min": round(min(rep_ms), 4), "ms_per_step_median": round(float(np.median(rep_ms)), 4),
            "ms_per_step_max": round(max(rep_ms), 4), "ms_per_step_repeats": len(rep_ms),
            "roofline": roofline,
        }
        line.update(extra)
        ws_, sl_ = model.config.world_size, model.shard_loss
        model.config.world_size, model.shard_loss = 1, False       # rank 0 alone: the gate is a single-rank evaluation
        try:
            line["parity"] = parity_gates(model, dev)
        finally:
            model.config.world_size, model.shard_loss = ws_, sl_
        if world == 1 and not args.no_cpu_baseline and args.config == 1:
            line["cpu_baseline"] = cpu_baseline()
    if world > 1 and interleaved and not args.no_sync_probe and not args.no_graph:
        # The second field is built LAST, behind a timer of its own: should its collectives hang on some box, rank 0 still prints
        # the line it has (with the reason in `sync_step`) and every rank leaves with status 0 -- the headline never depends on it.
        import threading

        def _sync_timeout():
            if rank == 0:
                line["sync_step"] = {"error": "no result after 240 s (the synchronous sharded step's probe was abandoned)"}
                print(json.dumps(line), flush=True)
            os._exit(0)
        guard = threading.Timer(240.0, _sync_timeout)
        guard.daemon = True
        guard.start()
        sync_probe()
        guard.cancel()
        if rank == 0:
            line.update(out_sync)
    if rank == 0:
        print(json.dumps(line))
    if watchdog is not None:
        watchdog.cancel()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
