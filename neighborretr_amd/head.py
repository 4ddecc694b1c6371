"""Orchestration of the HIP kernels for one step of the NeighborRetr loss head.

`head_losses(...)` is the fused equivalent of the reference's `_compute_losses`
(NeighborRetr/models/modeling.py:314-360) minus the token-clustering stage, whose outputs (the
global text/video tokens) come in as arguments:

    prepare (normalise+mask+bf16 split)  x4     modeling.py:495-496, :500-501
    token scorer MLP + masked softmax    x4     modeling.py:485-492
    fused local_level  BxB / BxM / MxB          modeling.py:499-512, :389-390
    global logits G (exact fp32 MFMA)           modeling.py:516-539
    Sinkhorn targets (both directions)          until_module.py:235-266
    centrality weights                          modeling.py:403-430
    row losses + finalize                       until_module.py:56-211, :285-289, :303-359

It is differentiable: `HeadLossFn` is a torch.autograd.Function whose backward runs the HIP
backward kernels (row losses, arg-max-routed similarity gradient, normalisation) and library
GEMMs for the scorer MLP.  Nothing here falls back to eager PyTorch for the forward math.
"""
import os

import torch

from . import hip, ops
from .capture_guard import record_event, wait_event, wait_stream

MLP_KEYS = ("0.weight", "0.bias", "2.weight", "2.bias")


class ScorerWeights:
    """bf16 hi/lo split of one token-scorer MLP (re-split whenever the parameters change)."""

    def __init__(self, w1, b1, w2, b2, want_lo=True):
        self.w1_hi, self.w1_lo = ops.split_bf16(w1.detach(), want_lo)
        self.b1 = b1.detach().float().contiguous()
        self.w2 = w2.detach().float().reshape(-1).contiguous()
        self.b2 = b2.detach().float().reshape(1).contiguous()
        self._w1 = w1.detach()
        self._w1t = None

    def w1_transposed(self):
        """bf16 pair of W1^T [d, H]: the operand of dX = dh W1 in the scorer's backward (built on first use per version)."""
        if self._w1t is None:
            from .cluster_fused import split_group
            H, d = self._w1.shape
            hi = torch.empty((d, H), dtype=torch.int16, device=self._w1.device)
            lo = torch.empty((d, H), dtype=torch.int16, device=self._w1.device)
            split_group([(self._w1.float().contiguous(), None, hi, lo, H, d, 1, H)])
            self._w1t = (hi, lo)
        return self._w1t


def token_weights(prep, mask, sw, n, N, prec, want_logits=False, scale_override=None):
    p = prep if scale_override is None else prep._replace(norm=scale_override)
    return ops.token_weights(p, sw.w1_hi, sw.w1_lo, sw.b1, sw.w2, sw.b2, mask, n, N, prec, want_logits)


def similarity_matrix(text_feat, video_feat, text_mask, video_mask, sw_t, sw_v, prec=hip.PREC_BF16X3):
    """local_level forward only (eval path): [A,Nt,d] x [Bv,Nv,d] -> S [A,Bv]."""
    A, Nt, _ = text_feat.shape
    Bv, Nv, _ = video_feat.shape
    lo = prec == hip.PREC_BF16X3
    pt = ops.prepare_tokens(text_feat, text_mask, want_lo=lo)
    pv = ops.prepare_tokens(video_feat, video_mask, want_lo=lo)
    w_t, _ = token_weights(pt, text_mask, sw_t, A, Nt, prec)
    w_v, _ = token_weights(pv, video_mask, sw_v, Bv, Nv, prec)
    S, _ = ops.local_level(pt, pv, w_t, w_v, A, Nt, Bv, Nv, prec)
    return S


def global_logits(gt, gv, sw_t1=None, sw_v1=None, keep=False):
    """global_level (modeling.py:516-539).  One global token per sample: an exact-fp32 GEMM.
    Several tokens (ActivityNet token counts): the fused kernel on un-normalised tokens, split-bf16.
    keep=True: returns (G, saved) with what the backward of the multi-token form needs (None for one token)."""
    B, Ngt, d = gt.shape
    Ngv = gv.shape[1]
    if Ngt == 1 and Ngv == 1:
        G = ops.gemm_nt_f32(gt.reshape(B, d), gv.reshape(gv.shape[0], d))
        return (G, None) if keep else G
    pt = ops.prepare_tokens(gt, None, normalize=False)
    pv = ops.prepare_tokens(gv, None, normalize=False)
    ones_t = torch.ones_like(pt.norm)
    ones_v = torch.ones_like(pv.norm)
    w_t, _ = token_weights(pt, None, sw_t1, B, Ngt, hip.PREC_BF16X3, scale_override=ones_t)
    w_v, _ = token_weights(pv, None, sw_v1, gv.shape[0], Ngv, hip.PREC_BF16X3, scale_override=ones_v)
    G, aux = ops.local_level(pt, pv, w_t, w_v, B, Ngt, gv.shape[0], Ngv, hip.PREC_BF16X3, hip.OUT_FULL, keep)
    return (G, dict(pt=pt, pv=pv, w_t=w_t, w_v=w_v, aux=aux)) if keep else G


def _check_global_tokens(gt, gv, hp):
    """More than one global token per sample (ActivityNet token counts: 64 -> 11 -> 3 text, 64 -> 16 -> 6 video
    tokens): the reference's centrality term multiplies [B] by [B,G] and raises (until_module.py:321), so there is
    no reference answer.  config.centrality_multi_token = "raise" (default) mirrors that; "mean" runs the step with
    w_i = mean over the sample's global tokens of the reference's per-token weight."""
    if gt.shape[1] == 1 and gv.shape[1] == 1:
        return
    mode = hp.get("centrality_multi_token", "raise")
    if mode == "mean":
        return
    if mode != "raise":
        raise ValueError(f"centrality_multi_token={mode!r}: expected 'raise' or 'mean'")
    raise RuntimeError(f"{gt.shape[1]} / {gv.shape[1]} global tokens per sample: the reference's centrality term fails to "
                       "broadcast at this shape (until_module.py:321); set config.centrality_multi_token='mean' to run "
                       "with the documented reduction (parity unpinned for that one term)")


PREC_MIXED = 2   # host-level plan: split-bf16 where logit_scale amplifies the error, bf16 elsewhere


def precision_plan(prec):
    """(batch x batch similarity, batch-token scorer, memory-bank paths) kernel precisions.

    The centrality term feeds S*logit_scale (x100) into a log-softmax, so single-pass bf16 error on
    the B x B similarity (~1.5e-4) shows up as ~2e-3 on that loss at small B.  The mixed plan keeps
    the B x B product (11% of the contraction flops at the MSR-VTT shape) in split-bf16 and runs the
    two memory-bank products -- whose outputs are only consumed as means over M entries -- and the
    bank-side scorer in one bf16 pass."""
    if prec == PREC_MIXED:
        return hip.PREC_BF16X3, hip.PREC_BF16X3, hip.PREC_BF16
    return prec, prec, prec


# The step's two bank products as chained tile pairs (one launch, nr_sim_pair_kernel).  OFF: measured alone the pair is
# 1.9 us shorter than two launches (33.8 -> 31.9 us), but the step with it is 2.5 % SLOWER (0.3308 vs 0.3227 ms, six A/B
# pairs of 1000 steps in one session): the first bank product can no longer start before the second chain's scorer has
# finished, and one 32 us launch that owns every CU's LDS holds the clustering kernels up longer than two 16 us ones.
# NR_PAIR_BANK=1 turns it on (developer A/B switch, tools/).
PAIR_BANK_PRODUCTS = os.environ.get("NR_PAIR_BANK", "0") == "1"
# Pipelined steps: the next step's bank chains wait for this step's row losses (see after_previous_push).  NR_TAIL_EDGE=0 turns
# the edge off (developer A/B switch, tools/ab_tail.sh).
TAIL_BEFORE_NEXT_BANK_READS = os.environ.get("NR_TAIL_EDGE", "1") == "1"
# NR_TAIL_EDGE=2 (A/B hook): the edge goes to the next step's bank PUSH instead (the row losses then only have to end before
# the next push, beside the next step's bank chain instead of in front of it)
TAIL_BEFORE_NEXT_PUSH = os.environ.get("NR_TAIL_EDGE", "1") == "2"
# Pipelined steps: the batch half of the local branch forked from the origin stream IN FRONT of the step's prologue
# (StepPipeline.early_fork), i.e. beside the previous step's bank chain instead of behind its push.  OFF: bit-identical, and
# 36 % SLOWER (0.354 vs 0.260 ms per step, two A/B pairs in one session, tools/ab_early_fork.sh) -- the same figure round 4 got
# from --decouple_push: whenever a step's chip-filling launches are released ahead of the previous step's push, the replayed
# graph runs slower, whatever the edges say.  NR_EARLY_FORK=1 turns it on (developer A/B switch).
EARLY_LOCAL_FORK = os.environ.get("NR_EARLY_FORK", "0") == "1"
# Loss-only split tail: the centrality weights computed inside the final row-loss launch (same arithmetic, same bits).
# NR_FUSE_CW=0: their own launch in front of it (developer A/B switch).
FUSE_CENTRALITY_WEIGHTS = os.environ.get("NR_FUSE_CW", "1") == "1"
# The token means on the second bank stream instead of the local branch: measured SLOWER (3773-3789 vs 3844-3861 steps/s,
# tools/ab_tail_fuse.sh; with the centrality weights fused: 3828-3839 vs 3873-3894) -- off; NR_COLSUM_OFF=1 turns it on (A/B switch)
COLSUM_OFF_CHAIN = os.environ.get("NR_COLSUM_OFF", "0") == "1"
# Loss-only split tail: the step's four token sets scored by ONE launch (nr_token_weights_fwd_group).  Alone that launch takes
# 51 us against 68 for the four (tools/microbench.py mlp) -- and the STEP is slower with it: 3773-3806 vs 3871-3888 steps/s at
# configs[1], 521 vs 562 at configs[3] (tools/ab_group_scorers.sh, two A/B pairs each): its 8-wave 192 x 256 workgroups (165
# registers a lane) leave no room on a CU for the clustering's workgroups, which then queue behind a 59 us launch (front kernel
# 40 us instead of 13 in the trace).  OFF; NR_GROUP_SCORERS=1 turns it on (developer A/B switch)
GROUP_SCORERS = os.environ.get("NR_GROUP_SCORERS", "0") == "1"
# Pipelined steps: the global logits on the tail stream (in front of the Sinkhorn solve) instead of closing the origin's part of
# the step.  OFF: bit-identical and 3.5 % SLOWER (3672-3689 vs 3807-3814 steps/s, three A/B pairs, tools/ab_logits_tail.sh) -- one
# more case of a shorter chain and a slower graph.  NR_LOGITS_TAIL=1 turns it on (developer A/B switch)
LOGITS_ON_TAIL = os.environ.get("NR_LOGITS_TAIL", "0") == "1"
# Loss-only step: the batch's text and video scorers as one launch (nr_token_weights_fwd_pair).  NR_PAIR_SCORERS=0: two launches (A/B).
PAIR_BATCH_SCORERS = os.environ.get("NR_PAIR_SCORERS", "1") == "1"
# ... from this many tokens in the smaller set on (a few workgroups per CU): configs[3] 517 -> 530 steps/s, configs[2] 424 -> 427;
# at configs[1] (3072 + 1536 tokens: one workgroup per CU) the two launches one after the other are faster, 3725 vs 3605
PAIR_BATCH_SCORERS_FROM = int(os.environ.get("NR_PAIR_SCORERS_FROM", "8192"))


def head_forward(text_feat, video_feat, text_mask, video_mask, mb_feat_t, mb_feat_v, mb_mask_t, mb_mask_v,
                 gt, gv, sw_t, sw_v, hp, logit_scale, prec=hip.PREC_BF16, keep=False, sw_t1=None, sw_v1=None, join=None,
                 bank_streams=None, local_stream=None, bank_early=0,
                 capture_order=((7, 5), (7, 1 << 30)), bank_prepared=None, prepared_out=None, bank_push=None, bb_late=False, slot=0,
                 pipeline=None, local_masks=None):
    """Forward of the head.  Returns (losses[5] device tensor, saved-state dict or None).

    `join`: optional callable run right before the first use of gt / gv.  Either the caller produces the
    global tokens on side streams while the local branch runs here and joins those streams in it, or --
    with gt = gv = None -- `join` runs the token clustering on the current stream and returns (gt, gv), while
    the local branch (prepare, scorer, B x B product) runs on `local_stream`; if `join` is a GENERATOR that
    issues one launch per next() and returns (gt, gv), its launches are interleaved with the local branch's.
    `bank_streams`: optional pair of side streams for the two memory-bank chains (see below).
    `bank_prepared`: optional (text, video) ops.Prepared of the bank (its persistent normalised bf16 shadow): the
    two bank prepare launches are skipped.  `prepared_out`: optional dict that receives the batch's prepared
    tokens ("pt", "pv") -- what the bank shadow is extended with at the push.
    `bank_push`: optional callable that pushes the batch into the memory bank (modeling.py:309-310).  The split tail
    calls it on its side stream as soon as both bank products have read the bank -- beside the Sinkhorn solve instead
    of behind it on the critical path; the other paths leave it to the caller.
    `pipeline` (split tail only; modeling.StepPipeline): consecutive steps captured overlapped.  The Sinkhorn solve then runs
    on `pipeline.tail_stream` (forked from this stream behind the logits) and NOTHING is joined back here: the streams still
    at work when this function returns -- solve, row losses, bank push -- are appended to `pipeline.pending`, to be joined
    into the capture's origin stream by whoever captures the steps (every join of a capture goes into its origin:
    capture_guard).  This stream is then free for the next step's prologue and clustering while this step's tail runs."""
    B, Nt, d = text_feat.shape
    Nv = video_feat.shape[1]
    M = mb_feat_v.shape[0]
    K = int(hp["num_neighbors"])
    if K > B:
        raise ValueError(f"num_neighbors={K} > batch={B}: the reference raises IndexError here "
                         "(until_module.py:119-123)")
    if M == 0 or mb_feat_t.shape[0] != M:
        raise ValueError("empty or inconsistent memory bank")
    if gt is None and join is None:
        raise ValueError("gt / gv missing and no join callable to produce them")
    p_bb, p_mlp, p_bank = precision_plan(prec)
    # masks as fp32 once (the loaders hand over int64); the kernels read them as multipliers
    text_mask, video_mask, mb_mask_t, mb_mask_v = (m if m.dtype == torch.float32 else m.float()
                                                   for m in (text_mask, video_mask, mb_mask_t, mb_mask_v))
    lo_b = keep or hip.PREC_BF16X3 in (p_bb, p_mlp, p_bank)
    lo_k = keep or p_bank == hip.PREC_BF16X3
    cur = torch.cuda.current_stream()
    # Loss-only step at B <= 128 ("split tail", below): the bank centralities stay PARTIAL sums (the row-loss kernel adds
    # them up itself: two reduction launches less per step)
    split_tail = (not keep) and bank_streams is not None and local_stream is not None and B <= 128 and B % 4 == 0

    # split tail: the two bank products as ONE launch of chained tile pairs (nr_local_level_group -> nr_sim_pair_kernel:
    # every workgroup computes a tile of the first and then a tile of the second product through one K loop) when both
    # run the 192 x 384 bf16 blocks on equally many tiles
    pair_bank = (split_tail and bank_early in (0, 2) and PAIR_BANK_PRODUCTS
                 and hip.local_level_group_kind(B, Nt, M, Nv, d, p_bank) == 0
                 and hip.local_level_group_kind(M, Nt, B, Nv, d, p_bank) == 0)

    # The local branch, one kernel launch per step (a generator, so that it can be interleaved launch by launch
    # with the clustering -- see below); its results land in `L`.
    L = {}
    early = [None, None]
    # Pipelined steps with an early fork (modeling.StepPipeline.early_fork): the local branch starts from a point of the origin
    # stream that lies BEFORE this step's prologue, so it converts its own copy of the caller's masks instead of reading the
    # prologue's (the same values: 0 / 1 as fp32)
    early_fork = (pipeline.early_fork if (pipeline is not None and EARLY_LOCAL_FORK and local_masks is not None and local_stream is not None
                                          and all(m is not None for m in local_masks)) else None)

    def fork_local():
        if early_fork is not None:
            wait_event(local_stream, early_fork)
        else:
            wait_stream(local_stream, cur)

    def local_steps():
        tm_l, vm_l = text_mask, video_mask
        if early_fork is not None:
            tm_l, vm_l = (m if m.dtype == torch.float32 else m.float() for m in local_masks)
            L["masks"] = (tm_l, vm_l)
        L["pt"], L["pv"] = pt_, pv_ = ops.prepare_tokens_pair(text_feat, tm_l, video_feat, vm_l, want_lo=lo_b, want_colsum=True)
        yield
        grouped = None
        if (GROUP_SCORERS and split_tail and bank_prepared is not None and bank_early > 1 and local_stream is not None
                and p_mlp == hip.PREC_BF16X3 and pt_.lo is not None):
            # ALL four token sets of the step in one scorer launch (nr_token_weights_fwd_group): the bank's sets read its prepared
            # shadow, so the launch waits for the previous step's push (as the step's prologue already did); the edge to the
            # previous step's row losses moves to the first bank product (bank_video_steps / bank_text_steps)
            if pipeline is not None and pipeline.prev_push_done is not None:
                wait_event(torch.cuda.current_stream(), pipeline.prev_push_done)
            pbt_, pbv_ = bank_prepared
            grouped = ops.token_weights_group(
                [(pt_, sw_t.w1_hi, sw_t.w1_lo, sw_t.b1, sw_t.w2, sw_t.b2, tm_l, B, Nt),
                 (pv_, sw_v.w1_hi, sw_v.w1_lo, sw_v.b1, sw_v.w2, sw_v.b2, vm_l, B, Nv),
                 (pbv_, sw_v.w1_hi, sw_v.w1_lo, sw_v.b1, sw_v.w2, sw_v.b2, mb_mask_v, M, Nv),
                 (pbt_, sw_t.w1_hi, sw_t.w1_lo, sw_t.b1, sw_t.w2, sw_t.b2, mb_mask_t, M, Nt)],
                [p_mlp, p_mlp, p_bank, p_bank])
        if grouped is not None:
            (L["w_t"], L["lg_t"]), (L["w_v"], L["lg_v"]), (L["w_bv"], _), (L["w_bt"], _) = grouped
            yield
        elif PAIR_BATCH_SCORERS and not keep and B * min(Nt, Nv) >= PAIR_BATCH_SCORERS_FROM:
            # the text and the video scorer in one grid; bit-identical to the two launches
            (L["w_t"], L["lg_t"]), (L["w_v"], L["lg_v"]) = ops.token_weights_pair(
                [(pt_, sw_t.w1_hi, sw_t.w1_lo, sw_t.b1, sw_t.w2, sw_t.b2, tm_l, B, Nt),
                 (pv_, sw_v.w1_hi, sw_v.w1_lo, sw_v.b1, sw_v.w2, sw_v.b2, vm_l, B, Nv)], p_mlp)
            yield
        else:
            L["w_t"], L["lg_t"] = token_weights(pt_, tm_l, sw_t, B, Nt, p_mlp, keep)
            yield
            L["w_v"], L["lg_v"] = token_weights(pv_, vm_l, sw_v, B, Nv, p_mlp, keep)
            yield
        if not (bb_late and split_tail):
            L["S"], L["aux0"] = ops.local_level(pt_, pv_, L["w_t"], L["w_v"], B, Nt, B, Nv, p_bb, hip.OUT_FULL, keep)
            yield
        # mean of the (unmasked) normalised tokens for the centrality weights, both in one launch.  Only the tail reads it: in the
        # split tail it runs on the second bank stream (idle until then), off the chain prepare -> scorers -> products -> push
        # that paces the pipelined steps; else here on the local branch
        if COLSUM_OFF_CHAIN and split_tail and bank_early > 1:
            side2_ = bank_streams[1]
            wait_stream(side2_, torch.cuda.current_stream())
            with torch.cuda.stream(side2_):
                L["mean_t"], L["mean_v"] = ops.colsum_pair(pt_.colsum, 1.0 / pt_.n_tok, pv_.colsum, 1.0 / pv_.n_tok)
            for t_ in (pt_.colsum, pv_.colsum):
                t_.record_stream(side2_)
        else:
            L["mean_t"], L["mean_v"] = ops.colsum_pair(pt_.colsum, 1.0 / pt_.n_tok, pv_.colsum, 1.0 / pv_.n_tok)
            yield
        # `bank_early` chains (0..2) run right behind the batch products, i.e. beside the clustering; the rest is
        # forked after the join (beside the Sinkhorn solve)
        if local_stream is not None and bank_early > 0:
            early[0] = yield from bank_video_steps()
            if bank_early > 1:
                early[1] = yield from bank_text_steps()
                if pair_bank:                         # both chains stopped in front of their product: one launch for the two
                    (pbv_, w_bv_, lg_bv_, _, _), (pbt_, w_bt_, lg_bt_, _, _) = early
                    c1_, c0_ = ops.local_level_group([(L["pt"], pbv_, L["w_t"], w_bv_, B, Nt, M, Nv, p_bank, hip.OUT_ROWSUM),
                                                      (pbt_, L["pv"], w_bt_, L["w_v"], M, Nt, B, Nv, p_bank, hip.OUT_COLSUM)])
                    yield
                    early[0], early[1] = (pbv_, w_bv_, lg_bv_, None, c1_), (pbt_, w_bt_, lg_bt_, None, c0_)

    def after_previous_push():
        """Pipelined steps: whatever reads the memory bank waits for the PREVIOUS step's push (on the stream it runs on)."""
        if pipeline is not None and pipeline.prev_push_done is not None:
            wait_event(torch.cuda.current_stream(), pipeline.prev_push_done)
        if pipeline is not None and pipeline.prev_tail_done is not None and TAIL_BEFORE_NEXT_BANK_READS:
            # ... and for the previous step's ROW LOSSES.  No data flows along this edge.  It is there because of how the ROCm 7.2
            # runtime orders a graph's nodes: a node nothing depends on (the row-loss launch is the last of its step) is put at the
            # END of the graph's hardware queues -- the row losses of all ten steps of a pipelined graph ran one after the other
            # when everything else was done, 24 of every 298 us with 32 workgroups on the chip; with a dependent it is started
            # before that dependent.  The bank chain starts ~150 us after the previous solve: nothing waits in practice
            # (profiles/r04_step_timeline_pipelined.txt: 3350-3480 -> 3490-3560 steps/s, A/B per box).
            wait_event(torch.cuda.current_stream(), pipeline.prev_tail_done)
            pipeline.prev_tail_done = None

    def bank_video_steps():
        # text x bank-video, row mean  -> centrality of text j  (used by the v2t neighbour loss)
        after_previous_push()
        if bank_prepared is not None:
            pbv = bank_prepared[1]
        else:
            pbv = ops.prepare_tokens(mb_feat_v, mb_mask_v, want_lo=lo_k)
            yield
        if "w_bv" in L:                                   # (scored by the step's grouped scorer launch)
            w_bv, lg_bv = L["w_bv"], None
        else:
            w_bv, lg_bv = token_weights(pbv, mb_mask_v, sw_v, M, Nv, p_bank, keep)
            yield
        if pair_bank:
            return pbv, w_bv, lg_bv, None, None          # the product itself: one launch with the other chain's (below)
        p1, aux1 = ops.local_level(L["pt"], pbv, L["w_t"], w_bv, B, Nt, M, Nv, p_bank, hip.OUT_ROWSUM, keep)
        yield
        if split_tail:
            return pbv, w_bv, lg_bv, aux1, p1
        c1 = ops.reduce_parts(p1, 1.0 / M)
        yield
        return pbv, w_bv, lg_bv, aux1, c1

    def bank_text_steps():
        # bank-text x video, column mean -> centrality of video j (used by the t2v neighbour loss)
        after_previous_push()
        if bank_prepared is not None:
            pbt = bank_prepared[0]
        else:
            pbt = ops.prepare_tokens(mb_feat_t, mb_mask_t, want_lo=lo_k)
            yield
        if "w_bt" in L:
            w_bt, lg_bt = L["w_bt"], None
        else:
            w_bt, lg_bt = token_weights(pbt, mb_mask_t, sw_t, M, Nt, p_bank, keep)
            yield
        if pair_bank:
            return pbt, w_bt, lg_bt, None, None
        p0, aux2 = ops.local_level(pbt, L["pv"], w_bt, L["w_v"], M, Nt, B, Nv, p_bank, hip.OUT_COLSUM, keep)
        yield
        if split_tail:
            return pbt, w_bt, lg_bt, aux2, p0
        c0 = ops.reduce_parts(p0, 1.0 / M)
        yield
        return pbt, w_bt, lg_bt, aux2, c0

    def exhaust(gen):
        """Runs a step generator to its end; returns its return value."""
        while True:
            try:
                next(gen)
            except StopIteration as stop:
                return stop.value

    bank_video = lambda: exhaust(bank_video_steps())     # noqa: E731
    bank_text = lambda: exhaust(bank_text_steps())       # noqa: E731

    stepwise_join = join is not None and hasattr(join, "__next__")
    if local_stream is not None and stepwise_join:
        # A captured HIP graph starts its nodes in CAPTURE ORDER: a node on another queue does not start before
        # the nodes captured ahead of it have started (profiled step: a branch captured after 7 clustering nodes
        # began 120 us late and ended up the critical path).  So the two branches are captured interleaved:
        # `capture_order` = [(clustering launches, local launches), ...] per turn, the last pair repeating.
        fork_local()
        loc = local_steps()
        loc_alive, clu_alive = True, True
        produced = None
        turn = 0
        while loc_alive or clu_alive:
            n_clu, n_loc = capture_order[min(turn, len(capture_order) - 1)]
            turn += 1
            if clu_alive:
                try:
                    for _ in range(n_clu if loc_alive else 1 << 30):
                        next(join)
                except StopIteration as stop:
                    produced, clu_alive = stop.value, False
            if loc_alive:
                torch.cuda.set_stream(local_stream)
                try:
                    for _ in range(n_loc if clu_alive else 1 << 30):
                        next(loc)
                except StopIteration:
                    loc_alive = False
                finally:
                    torch.cuda.set_stream(cur)
        if produced is not None:
            gt, gv = produced
    else:
        # `join`: one callable run right before gt / gv are needed (may return them); generators are run through
        if local_stream is not None:
            fork_local()
            if join is not None:
                produced = exhaust(join) if stepwise_join else join()
                if produced is not None:
                    gt, gv = produced
                join = None
            torch.cuda.set_stream(local_stream)
        try:
            exhaust(local_steps())
        finally:
            if local_stream is not None:
                torch.cuda.set_stream(cur)
        if join is not None:
            produced = exhaust(join) if stepwise_join else join()
            if produced is not None:
                gt, gv = produced
    pt, pv, w_t, w_v, lg_t, lg_v = L["pt"], L["pv"], L["w_t"], L["w_v"], L["lg_t"], L["lg_v"]
    if prepared_out is not None:
        prepared_out["pt"], prepared_out["pv"] = pt, pv
    S, aux0, mean_t, mean_v = L.get("S"), L.get("aux0"), L["mean_t"], L["mean_v"]
    _check_global_tokens(gt, gv, hp)
    gt2 = gt.float().contiguous()                  # [B, G, d]: G = 1 at the MSR-VTT token counts
    gv2 = gv.float().contiguous()
    ls = logit_scale.detach().float().reshape(1).contiguous()
    # Loss-only step at B <= 128 ("split tail"): the global logits and the Sinkhorn solve depend on the clustering
    # alone, so they follow it on THIS stream without waiting for the local branch; the Sinkhorn kernel emits the
    # uniform-CE row terms itself, and everything that needs the local branch (leftover bank chains, bank push,
    # centrality weights, the row-loss kernel with its top-K) gathers on a side stream.  The two tail kernels finalize
    # themselves (whichever workgroup finishes last reduces the row terms to the five losses), so the critical path
    # is prologue -> clustering -> logits -> Sinkhorn, with neither a finalize launch nor the push behind it.
    if split_tail:
        tail = pipeline.tail_stream if pipeline is not None else None
        if tail is not None and LOGITS_ON_TAIL:
            # pipelined steps: the global logits (one small launch) open the TAIL stream instead of closing this one -- the next
            # step's prologue follows the clustering's last launch directly
            wait_stream(tail, cur)
            with torch.cuda.stream(tail):
                G = global_logits(gt, gv, sw_t1, sw_v1)
                g_ready = record_event(tail)
            for t_ in (gt, gv):
                t_.record_stream(tail)
        else:
            G = global_logits(gt, gv, sw_t1, sw_v1)
            g_ready = record_event(cur)
        rowloss = torch.empty((2, 4, B), dtype=torch.float32, device=G.device)
        losses = torch.empty((5,), dtype=torch.float32, device=G.device)
        counter = ops.split_tail_counter(G.device, slot)
        wts = (hp["uniform_weight"], hp["neighbor_weight"], hp["kl_weight"])
        # The two self-finalizing launches share `counter`; only the one that finishes last resets it.  If anything raises
        # between them (the push callable, an NR_E* status), the word would stay non-zero and every later step would finalize
        # on incomplete row terms without an error: zero it before passing the exception on.
        try:
            if tail is not None:
                wait_stream(tail, cur)
                with torch.cuda.stream(tail):
                    ops.sinkhorn_uniform_rows_final(G, hp["beta"], hp["temperature"], rowloss, counter, *wts, losses, 50)
                for t_ in (G, rowloss, losses):
                    t_.record_stream(tail)
            else:
                ops.sinkhorn_uniform_rows_final(G, hp["beta"], hp["temperature"], rowloss, counter, *wts, losses, 50)
            tgt_r = tgt_c = None
            side, side2 = bank_streams[0], bank_streams[1]
            push_stream = bank_streams[2] if len(bank_streams) > 2 else None
            wait_stream(side, local_stream)
            wait_stream(side2, local_stream)
            with torch.cuda.stream(side2):
                pbt, w_bt, lg_bt, aux2, c0 = early[1] or bank_text()
            with torch.cuda.stream(side):
                if S is None:
                    # `bb_late`: the batch x batch product runs HERE, beside the Sinkhorn solve (only the row losses read S)
                    # -- one chip-filling launch less beside the clustering
                    S, aux0 = ops.local_level(pt, pv, w_t, w_v, B, Nt, B, Nv, p_bb, hip.OUT_FULL, keep)
                pbv, w_bv, lg_bv, aux1, c1 = early[0] or bank_video()
                wait_stream(side, side2)
                if pair_bank and early[0] is None:
                    for t_ in (pbt.hi, w_bt):
                        t_.record_stream(side)
                    c1, c0 = ops.local_level_group([(L["pt"], pbv, L["w_t"], w_bv, B, Nt, M, Nv, p_bank, hip.OUT_ROWSUM),
                                                    (pbt, L["pv"], w_bt, L["w_v"], M, Nt, B, Nv, p_bank, hip.OUT_COLSUM)])
                if bank_push is not None:
                    # both bank products have read the bank (and its prepared shadow): the batch may take the oldest rows'
                    # place -- on a stream of its own, beside the centrality weights and the row losses.  (A stream that
                    # has already been joined must not be forked again inside one capture: putting the push back on
                    # `side2` after `wait_stream(side, side2)` made the ROCm 7.2 runtime segfault at capture time.)
                    pst = push_stream if push_stream is not None else side
                    if push_stream is not None:
                        wait_stream(push_stream, side)
                    with torch.cuda.stream(pst):
                        for t_ in (pt.hi, pt.norm, pv.hi, pv.norm) + ((pt.lo, pv.lo) if pt.lo is not None else ()):
                            t_.record_stream(pst)
                        if early_fork is not None and pipeline.prologue_done is not None:
                            wait_event(pst, pipeline.prologue_done)      # the push writes at the ring head this step's prologue has moved
                        if pipeline is not None and pipeline.prev_tail_done is not None and TAIL_BEFORE_NEXT_PUSH:
                            wait_event(pst, pipeline.prev_tail_done)
                            pipeline.prev_tail_done = None
                        with torch.no_grad():
                            bank_push()
                wait_event(side, g_ready)
                if FUSE_CENTRALITY_WEIGHTS and gt2.shape[1] == 1 and gv2.shape[1] == 1 and gt2.shape[0] == B and gv2.shape[0] == B:
                    # one global token per sample: the row-loss launch computes the centrality weights itself (one launch less
                    # on the chain that the next step's bank reads wait for)
                    wc_t = wc_v = cw_aux = None
                    ops.row_losses_no_uniform_final_cw(S, G, c0, c1, 1.0 / M, gt2.view(B, -1), gv2.view(B, -1), mean_t, mean_v,
                                                       hp["centrality_scale"], ls, K, hp["temperature"], rowloss, counter, *wts, losses)
                else:
                    wc_t, wc_v, cw_aux = ops.centrality_weights_pair(gt2, gv2, mean_t, mean_v, hp["centrality_scale"], keep)
                    ops.row_losses_no_uniform_final(S, G, c0, c1, 1.0 / M, wc_t, wc_v, ls, K, hp["temperature"], rowloss, counter,
                                                    *wts, losses)
                if pipeline is not None:
                    pipeline.tail_done = record_event(side)
        except BaseException:
            if not torch.cuda.is_current_stream_capturing():
                torch.cuda.synchronize(G.device)
                counter.zero_()
            else:
                ops.forget_split_tail_counter(G.device, slot)   # an aborted capture: the next step takes a fresh zeroed word
            raise
        if pipeline is not None:
            # everything this step allocated stays referenced until the capture ends (StepPipeline.own): its forked streams are
            # not joined here, and the next step allocates on this stream right away
            pipeline.own(L, early, gt, gv, gt2, gv2, ls, G, rowloss, losses, S, aux0, c0, c1, wc_t, wc_v, cw_aux,
                         pbt, w_bt, lg_bt, aux2, pbv, w_bv, lg_bv, aux1, text_mask, video_mask, mb_mask_t, mb_mask_v)
            pipeline.pending += [tail, side] + ([push_stream] if (push_stream is not None and bank_push is not None) else [])
            # Nothing is joined into this stream, which goes straight on to the NEXT step: whatever was allocated on it and is
            # still read by this step's forked streams must not go back to its free list when this function returns
            for t_ in (gt2, gv2, gt, gv, ls, logit_scale, text_mask, video_mask, G, rowloss, losses):
                if torch.is_tensor(t_) and t_.is_cuda:
                    for st_ in (tail, side, side2, local_stream) + ((push_stream,) if (push_stream is not None and bank_push is not None) else ()):
                        t_.record_stream(st_)
        else:
            wait_stream(cur, side)
            if push_stream is not None:
                wait_stream(cur, push_stream)
        for t_ in (S, mean_t, mean_v, w_t, w_v, pt.hi, pv.hi, c0, c1, wc_t, wc_v, G, rowloss, losses):
            if t_ is None:                   # (the centrality weights: computed inside the row-loss launch)
                continue
            t_.record_stream(cur)
            t_.record_stream(side)
        if pt.lo is not None:
            pt.lo.record_stream(cur)
            pv.lo.record_stream(cur)
        pt.norm.record_stream(cur)
        pv.norm.record_stream(cur)
    else:
        # The critical path after the join is global logits -> Sinkhorn -> row losses: captured FIRST.  The logits and the solve
        # need the global tokens only, so this stream joins the local branch (whose tail, with the bank chains run early, is
        # the last bank product) BEHIND the solve, not in front of it.  What is left of the bank chains and the centrality
        # weights feed only the row-loss kernel; with `bank_streams` they are forked from the join point (an event recorded
        # before the Sinkhorn launch) and run beside the solve.
        fork = None
        if bank_streams is not None:
            fork = record_event(cur)
        G, g_saved = global_logits(gt, gv, sw_t1, sw_v1, keep=True) if keep else (global_logits(gt, gv, sw_t1, sw_v1), None)
        tgt_r, tgt_c = ops.sinkhorn_targets(G, hp["beta"], 50)
        if local_stream is not None:
            wait_stream(cur, local_stream)
            for t_ in (S, mean_t, mean_v, w_t, w_v, pt.hi, pv.hi):
                t_.record_stream(cur)
        if bank_streams is not None:
            for st_ in bank_streams:
                wait_event(st_, fork)
                if local_stream is not None:
                    wait_stream(st_, local_stream)       # (their launches read the local branch's tokens / weights / means)
            with torch.cuda.stream(bank_streams[1]):
                pbt, w_bt, lg_bt, aux2, c0 = early[1] or bank_text()
            with torch.cuda.stream(bank_streams[0]):
                pbv, w_bv, lg_bv, aux1, c1 = early[0] or bank_video()
                wc_t, wc_v, cw_aux = ops.centrality_weights_pair(gt2, gv2, mean_t, mean_v, hp["centrality_scale"], keep)
            for st_ in bank_streams:
                wait_stream(cur, st_)
            for t_ in (c0, c1, wc_t, wc_v) + ((pbt.hi, pbv.hi, w_bt, w_bv) if keep else ()):
                t_.record_stream(cur)
        else:
            pbv, w_bv, lg_bv, aux1, c1 = early[0] or bank_video()
            pbt, w_bt, lg_bt, aux2, c0 = early[1] or bank_text()
            wc_t, wc_v, cw_aux = ops.centrality_weights_pair(gt2, gv2, mean_t, mean_v, hp["centrality_scale"], keep)
        rowloss, losses = ops.row_losses_final(S, G, tgt_r, tgt_c, c0, c1, wc_t, wc_v, ls, K, hp["temperature"],
                                               hp["uniform_weight"], hp["neighbor_weight"], hp["kl_weight"])
    saved = None
    if keep:
        saved = dict(pt=pt, pv=pv, pbt=pbt, pbv=pbv, w_t=w_t, w_v=w_v, w_bt=w_bt, w_bv=w_bv,
                     lg_t=lg_t, lg_v=lg_v, lg_bt=lg_bt, lg_bv=lg_bv,
                     aux=(aux0, aux1, aux2), S=S, G=G, tgt_r=tgt_r, tgt_c=tgt_c,
                     c0=c0, c1=c1, wc_t=wc_t, wc_v=wc_v, cw_aux=cw_aux, mean_t=mean_t, mean_v=mean_v,
                     ls=ls, gt2=gt2, gv2=gv2, g_saved=g_saved)
    return losses, saved


def _rows(prep, first, count):
    """Rows [first, first + count) of a Prepared token set (contiguous views)."""
    return ops.Prepared(prep.hi[first:first + count], prep.lo[first:first + count] if prep.lo is not None else None,
                        prep.norm[first:first + count], None, count, prep.d)


def head_forward_sharded(text_feat, video_feat, text_mask, video_mask, mb_feat_t, mb_feat_v, mb_mask_t, mb_mask_v,
                         gt, gv, sw_t, sw_v, hp, logit_scale, prec, rank, world, bank_prepared=None, prepared_out=None,
                         sw_t1=None, sw_v1=None, join=None, local_stream=None):
    """Loss-only forward with the similarity and bank work SHARDED over `world` ranks: rank r owns the
    samples [r*b, (r+1)*b) of the gathered batch and computes the two slabs of S it needs for its rows of either
    direction (2/W of the batch x batch product), its slice of both bank centrality vectors (1/W of the bank
    products) and its rows of the row losses.  Exchanged: the centrality slices (one all-gather of 2b floats) and
    the row terms (one all-reduce of 8B floats); the token scorers, the global logits and the Sinkhorn solve are
    small and stay replicated.  Returns the same [5] losses on every rank as head_forward.

    `join` (gt = gv = None): callable that produces the global tokens -- the rank's share of the clustering plus the
    all-gather of the [b, c, d] tokens -- on the CURRENT stream, while the local branch (prepare, scorers, the four
    products) runs on `local_stream`; the global logits and the Sinkhorn solve follow the clustering without waiting for
    the local branch.  Both collectives of this function are issued on the current stream, in program order, so that
    every rank enqueues the same sequence whatever its streams do.  No host synchronisation anywhere: with the "nccl"
    backend (RCCL) the whole sharded step captures into ONE HIP graph (tools/rccl_capture_probe.py,
    profiles/r03_rccl_capture.txt)."""
    from . import comm
    B, Nt, d = text_feat.shape
    Nv = video_feat.shape[1]
    M = mb_feat_v.shape[0]
    K = int(hp["num_neighbors"])
    if B % world:
        raise ValueError("the gathered batch must divide over the ranks")
    if K > B:
        raise ValueError(f"num_neighbors={K} > batch={B}")
    if gt is None and join is None:
        raise ValueError("gt / gv missing and no join callable to produce them")
    b, r0 = B // world, rank * (B // world)
    p_bb, p_mlp, p_bank = precision_plan(prec)
    text_mask, video_mask, mb_mask_t, mb_mask_v = (m if m.dtype == torch.float32 else m.float()
                                                   for m in (text_mask, video_mask, mb_mask_t, mb_mask_v))
    lo_b = hip.PREC_BF16X3 in (p_bb, p_mlp, p_bank)
    L = {}

    def local_branch():
        pt, pv = ops.prepare_tokens_pair(text_feat, text_mask, video_feat, video_mask, want_lo=lo_b, want_colsum=True)
        w_t, _ = token_weights(pt, text_mask, sw_t, B, Nt, p_mlp)
        w_v, _ = token_weights(pv, video_mask, sw_v, B, Nv, p_mlp)
        if prepared_out is not None:
            prepared_out["pt"], prepared_out["pv"] = pt, pv
        pt_r, pv_r = _rows(pt, r0 * Nt, b * Nt), _rows(pv, r0 * Nv, b * Nv)
        w_t_r, w_v_r = w_t[r0:r0 + b].contiguous(), w_v[r0:r0 + b].contiguous()
        L["S_rows"], _ = ops.local_level(pt_r, pv, w_t_r, w_v, b, Nt, B, Nv, p_bb, hip.OUT_FULL)          # S[r0:r0+b, :]
        L["S_cols"], _ = ops.local_level(pt, pv_r, w_t, w_v_r, B, Nt, b, Nv, p_bb, hip.OUT_FULL)          # S[:, r0:r0+b]
        if bank_prepared is not None:
            pbt, pbv = bank_prepared
        else:
            lo_k = p_bank == hip.PREC_BF16X3
            pbt = ops.prepare_tokens(mb_feat_t, mb_mask_t, want_lo=lo_k)
            pbv = ops.prepare_tokens(mb_feat_v, mb_mask_v, want_lo=lo_k)
        w_bt, _ = token_weights(pbt, mb_mask_t, sw_t, M, Nt, p_bank)
        w_bv, _ = token_weights(pbv, mb_mask_v, sw_v, M, Nv, p_bank)
        p1, _ = ops.local_level(pt_r, pbv, w_t_r, w_bv, b, Nt, M, Nv, p_bank, hip.OUT_ROWSUM)
        p0, _ = ops.local_level(pbt, pv_r, w_bt, w_v_r, M, Nt, b, Nv, p_bank, hip.OUT_COLSUM)
        L["mine"] = torch.stack((ops.reduce_parts(p0, 1.0 / M), ops.reduce_parts(p1, 1.0 / M)))          # [2, b]
        L["mean_t"], L["mean_v"] = ops.colsum_pair(pt.colsum, 1.0 / pt.n_tok, pv.colsum, 1.0 / pv.n_tok)
        L["keep"] = (pt, pv, w_t, w_v, pbt, pbv, w_bt, w_bv, pt_r, pv_r, w_t_r, w_v_r)                       # alive until the join

    cur = torch.cuda.current_stream()
    two = local_stream is not None and join is not None
    if two:
        wait_stream(local_stream, cur)
        with torch.cuda.stream(local_stream):
            local_branch()
    if join is not None:
        gt, gv = join()
    if not two:
        local_branch()
    _check_global_tokens(gt, gv, hp)
    gt2 = gt.float().contiguous()
    gv2 = gv.float().contiguous()
    G = global_logits(gt, gv, sw_t1, sw_v1)
    tgt_r, tgt_c = ops.sinkhorn_targets(G, hp["beta"], 50)
    if two:
        wait_stream(cur, local_stream)
        used_here = [L["S_rows"], L["S_cols"], L["mine"], L["mean_t"], L["mean_v"]]
        for prep in L["keep"][:2]:                   # the batch's prepared tokens: the bank push (caller, this stream) reads them
            used_here += [t_ for t_ in (prep.hi, prep.lo, prep.norm) if t_ is not None]
        for t_ in used_here:
            t_.record_stream(cur)
    mine = L["mine"]
    everyone = torch.empty((world, 2, b), dtype=torch.float32, device=mine.device)
    comm.all_gather_into_tensor(everyone.view(-1), mine.view(-1))
    c0 = everyone[:, 0, :].reshape(B).contiguous()
    c1 = everyone[:, 1, :].reshape(B).contiguous()
    wc_t, wc_v, _ = ops.centrality_weights_pair(gt2, gv2, L["mean_t"], L["mean_v"], hp["centrality_scale"], False)
    ls = logit_scale.detach().float().reshape(1).contiguous()
    rowloss = ops.row_losses_slab(L["S_rows"], L["S_cols"], r0, G, tgt_r, tgt_c, c0, c1, wc_t, wc_v, ls, K, hp["temperature"])
    comm.all_reduce(rowloss)                       # every row was written by exactly one rank, zeros elsewhere
    return ops.loss_finalize(rowloss, hp["uniform_weight"], hp["neighbor_weight"], hp["kl_weight"])
