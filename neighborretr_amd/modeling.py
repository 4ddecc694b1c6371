"""Host-side mirror of the reference's model API for the similarity / loss path.

`NeighborRetr` keeps the constructor, method names, argument order, return tuples, plain
`mb_*` attributes and state-dict parameter names of NeighborRetr/models/modeling.py:46-658, so
the reference's trainer / evaluator / MemoryBankManager can drive it unchanged, while every
method on the hot path dispatches to the HIP kernels (neighborretr_amd.head / .ops).

Out of scope here: the CLIP towers and the temporal transformer.  They are
feature PRODUCERS; pass any module with `encode_text` / `encode_image` + `logit_scale` as `clip=`
(e.g. the reference's own CLIP on PyTorch-ROCm).  Without one the model runs in feature mode:
`text_ids` / `video` given to forward() are taken to be the token features [b,Nt,d] / [b,Nv,d]
already -- which is what bench.py, the tests and `main_retrieval.py --synthetic` use.
"""
import contextlib
import math
import os
from types import SimpleNamespace

import torch
from torch import nn

from . import head, hip, ops
from .capture_guard import wait_stream
from .cluster import CTM, TCBlock
from .until_module import (AllGather, CentralityWeightingLoss, KLDivergenceLoss, NeighborAdjustingLoss,
                           UniformRegularizationLoss)

allgather = AllGather.apply

DEFAULTS = dict(centrality_scale=0.3, beta=0.7, num_neighbors=20, temperature=3.0, uniform_weight=1.0,
                neighbor_weight=1.0, kl_weight=1.0, world_size=1, local_rank=0,   # args_parser.py:26-41
                # this build: what the centrality term does when a sample keeps SEVERAL global tokens (ActivityNet token
                # counts).  "raise" = the reference's behaviour (until_module.py:321 fails to broadcast); "mean" = weight
                # averaged over the sample's global tokens (head._check_global_tokens)
                centrality_multi_token="raise")


class FeatureModeCLIP(nn.Module):
    """Stand-in for the CLIP tower in feature mode: holds only `logit_scale` (CLIP's init, ln(1/0.07))."""

    def __init__(self):
        super().__init__()
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))


class StepPipeline:
    """Consecutive loss-only steps captured into ONE HIP graph so that they OVERLAP (bench.py --unroll).  Inside a capture every
    join has to go into the capture's ORIGIN stream (capture_guard: a forked stream that joins its own children takes the ROCm
    7.2 runtime down), so a whole step cannot run on a stream of its own.  Instead the origin carries only what is serial by
    nature -- prologue -> clustering -> global logits of step k, then of step k + 1, ... -- and everything else of a step is
    forked and NOT joined back until the capture ends: the local branch and the bank chains as before, and now also the
    Sinkhorn solve (`tail_stream`; two workgroups, 37 us) and with it the whole tail.  The one true dependency between two
    loss-only steps is the memory bank: step k + 1's prologue (it moves the ring head) and bank chains wait for step k's push
    (`push_done` / `prev_push_done`).  While one of
    these is installed as `model._pipeline`, a step (a) takes its side streams and its finalize word from `slot` (a stream
    forked in one step is never forked again in the same capture), (b) leaves `push_done` behind (event: the batch is in the
    bank and its prepared shadow) and (c) appends the streams it left at work to `pending`."""

    def __init__(self, slot, prev=None, decoupled=False):
        self.slot = int(slot)
        # decoupled: the NEXT step's prologue does not wait for this step's push (own copy of the ring head per step; only the
        # bank chains wait).  Correct (bit-identical), and SLOWER: 0.355 ms per step against 0.293 with the origin waiting for the
        # push before the next prologue -- the steps then overlap so far that both chains of two steps share the CUs
        # (bench.py --decouple_push; A/B in one session, round 4).  Off by default.
        self.decoupled = bool(decoupled)
        self.push_done = None
        self.prev_push_done = prev.push_done if prev is not None else None    # the bank reads of THIS step wait for it
        self.head = None             # this step's own copy of the ring head (its push reads it; the next prologue moves the shared one)
        self.tail_stream = None
        self.tail_done = None        # event: this step's five losses are written (head.head_forward, split tail)
        self.prev_tail_done = prev.tail_done if prev is not None else None
        self.pending = []
        # EARLY FORK of the local branch: an event recorded on the origin stream BEFORE it waits for the previous step's push
        # (set by whoever captures the steps).  The batch half of the local branch -- prepare, batch scorers, batch x batch
        # product -- reads neither the bank nor anything the prologue writes, so it forks from there and runs beside the previous
        # step's bank chain instead of behind its push; `prologue_done` (recorded behind this step's prologue) is what this
        # step's push then waits for explicitly (it writes at the ring head the prologue has just moved).
        self.early_fork = None
        self.prologue_done = None
        # OWNERSHIP: every tensor this step allocates that a forked stream (tail, bank chains, push) may still read when the origin
        # stream has moved on to the next step is referenced from here until the capture ends -- whoever captures the steps keeps
        # the StepPipeline objects alive that long.  A block that is never freed inside the capture is never handed out again
        # inside it, so no wait edge and no record_stream() call has to be right for a tail's operands to stay intact
        # (bench.py --decouple_push, whose steps run far ahead of each other, relied on exactly those).
        self.owned = []

    def own(self, *objs):
        """Takes ownership of every CUDA tensor reachable from `objs` (dicts, sequences, objects with fields)."""
        seen = set()
        stack = list(objs)
        while stack:
            o = stack.pop()
            if o is None or isinstance(o, (int, float, str, bool, bytes)) or id(o) in seen:
                continue
            seen.add(id(o))
            if torch.is_tensor(o):
                if o.is_cuda:
                    self.owned.append(o)
            elif isinstance(o, dict):
                stack.extend(o.values())
            elif isinstance(o, (list, tuple, set)):
                stack.extend(o)
            elif hasattr(o, "_fields"):                        # namedtuple-like (ops.Prepared)
                stack.extend(getattr(o, f) for f in o._fields)
            elif hasattr(o, "__dict__") and type(o).__module__.startswith("neighborretr_amd") and not isinstance(o, nn.Module):
                stack.extend(vars(o).values())


class OwnedSlot:
    """Persistent buffers of an OVERLAPPED owned step of the step-interleaved job (`model.interleave_overlap`): the gathered batch,
    and the memory bank as the step's loss has to see it -- the prepared bf16 shadow, the masks and the noise counter, copied
    before the batch is pushed.  With these the loss of step k reads nothing that the exchange-and-push steps k + 1, k + 2, ...
    write, so it runs on a stream of its own BESIDE them instead of in front of them.  Allocated outside
    any capture: two graphs (the exchange half and the loss half of the step) name these addresses."""

    def __init__(self, lay, shadow, masks, rng, copy_lo):
        from .ops import Prepared
        dev = rng.device
        B = lay["W"] * lay["b"]
        dt = (torch.float32, torch.float32, torch.int64, torch.float32, torch.float32)
        self.batch = [torch.empty((B,) + tuple(shp), dtype=t, device=dev) for shp, t in zip(lay["shapes"], dt)]
        self.copy_lo = bool(copy_lo)
        self.shadow = tuple(Prepared(torch.empty_like(p.hi), torch.empty_like(p.lo) if copy_lo else None, torch.empty_like(p.norm),
                                     None, p.n_tok, p.d) for p in shadow)
        self.masks = tuple(torch.empty_like(m) for m in masks)
        self.rng = torch.empty_like(rng)
        self.stream = torch.cuda.Stream(device=dev)      # the stream this slot's loss half runs on (one per slot: see `owned_slots`)
        self.scratch = {}                 # the loss half's `_last_prepared` (never the live model's)
        self.loss_done = None             # event: the loss half of the previous use of this slot has finished reading it
        self.losses = None

    def _pairs(self, shadow, masks, rng):
        dst = [self.shadow[0].hi, self.shadow[1].hi, self.shadow[0].norm, self.shadow[1].norm, self.masks[0], self.masks[1], self.rng]
        src = [shadow[0].hi, shadow[1].hi, shadow[0].norm, shadow[1].norm, masks[0], masks[1], rng]
        if self.copy_lo:
            dst += [self.shadow[0].lo, self.shadow[1].lo]
            src += [shadow[0].lo, shadow[1].lo]
        return dst, src

    def fits(self, lay, shadow, masks, copy_lo):
        return (self.copy_lo == bool(copy_lo) and all(tuple(a.shape[1:]) == tuple(s) for a, s in zip(self.batch, lay["shapes"]))
                and self.batch[0].shape[0] == lay["W"] * lay["b"]
                and all(a.hi.shape == b.hi.shape for a, b in zip(self.shadow, shadow))
                and all(a.shape == b.shape and a.dtype == b.dtype for a, b in zip(self.masks, masks)))

    def take(self, shadow, masks, rng):
        """The bank's prepared shadow, its masks and the noise counter -> this slot (device copies on the current stream)."""
        dst, src = self._pairs(shadow, masks, rng)
        ops.copy_group(dst, src)                  # one launch (nr_copy_group); torch._foreach_copy_ issues one copy kernel per tensor here


class NeighborRetr(nn.Module):
    def __init__(self, config, clip=None, width=512, precision="bf16", with_encoders=False, encoder_dims=None):
        """clip: any module with encode_text / encode_image / logit_scale (e.g. the reference's own CLIP).
        with_encoders=True: build the ViT-B/32 towers + the temporal transformer of the reference model here
        (neighborretr_amd.encoders: stock PyTorch-ROCm modules, reference parameter names, random init unless a
        checkpoint is loaded) -- BASELINE configs[4], pixel / token-id inputs.  Neither: feature mode."""
        super().__init__()
        self.config = config
        for k, v in DEFAULTS.items():
            if not hasattr(config, k):
                setattr(config, k, v)
        self.transformer_width = width
        if clip is None and with_encoders:
            from .encoders import ClipEncoders, TemporalTransformer
            clip = ClipEncoders(**(encoder_dims or {}))
            width = self.transformer_width = clip.transformer.width
            # modeling.py:154-166: frame position embeddings + `num_hidden_layers` temporal blocks
            self.frame_position_embeddings = nn.Embedding(clip.context_length, width)
            self.transformerClip = TemporalTransformer(width, int(getattr(config, "num_hidden_layers", 4)), width // 64)
        self.clip = clip if clip is not None else FeatureModeCLIP()
        self.feature_mode = clip is None
        self.encoder_dtype = torch.bfloat16
        # "bf16"    training plan: bank products / bank scorer one bf16 pass, B x B product split-bf16
        # "bf16x3"  everything split-bf16 (rank-exact eval, golden parity)
        # "bf16_all" everything one bf16 pass (fastest; ~2e-3 on the centrality loss at small B)
        self.precision = precision
        # token scorers -- all eight exist for checkpoint compatibility; only text/video_weight_fc
        # and *_fc1 are ever used (modeling.py:137-146)
        for name in ("text_weight_fc", "video_weight_fc", "text_weight_fc0", "video_weight_fc0",
                     "text_weight_fc1", "video_weight_fc1", "text_weight_intra", "video_weight_intra"):
            setattr(self, name, self._scorer(width))
        self.centrality_weighting_loss = CentralityWeightingLoss()
        self.neighbor_adjusting_loss = NeighborAdjustingLoss()
        self.uniform_regularization_loss = UniformRegularizationLoss()
        self.kl_loss = KLDivergenceLoss()
        self._init_memory_bank()
        clip_sd = None
        if hasattr(self, "transformerClip"):
            clip_sd = {k: v.clone() for k, v in self.clip.state_dict().items()}
        self.apply(self._init_weights)
        if clip_sd is not None:
            # modeling.py:62-69: the model-wide init runs over the CLIP towers too, then CLIP's own weights are put back
            # and the temporal transformer starts from the text tower's first blocks / positional embedding (:199-219)
            self.clip.load_state_dict(clip_sd)
            own = {"frame_position_embeddings.weight": clip_sd["positional_embedding"].clone()}
            for k, v in clip_sd.items():
                if k.startswith("transformer.resblocks.") and int(k.split(".")[2]) < len(self.transformerClip.resblocks):
                    own[k.replace("transformer.", "transformerClip.", 1)] = v.clone()
            self.load_state_dict(own, strict=False)
        # token clustering (modeling.py:186-197)
        self.text_ctm0 = CTM(sample_ratio=1 / 6, embed_dim=width, dim_out=width, k=3)
        self.text_block0 = TCBlock(dim=width, num_heads=8)
        self.text_ctm1 = CTM(sample_ratio=1 / 4, embed_dim=width, dim_out=width, k=3)
        self.text_block1 = TCBlock(dim=width, num_heads=8)
        self.video_ctm0 = CTM(sample_ratio=1 / 4, embed_dim=width, dim_out=width, k=3)
        self.video_block0 = TCBlock(dim=width, num_heads=8)
        self.video_ctm1 = CTM(sample_ratio=1 / 3, embed_dim=width, dim_out=width, k=3)
        self.video_block1 = TCBlock(dim=width, num_heads=8)
        self._scorer_cache = {}
        self._join_global = None
        self.cluster_side_stream = True      # training step: clustering (forward + backward) on its own stream, see _compute_losses
        self.interleave_training_forward = True   # ... its forward launches interleaved with the head's local branch
        # side streams inside the step (and inside its capture).  With ONE hardware queue (GPU_MAX_HW_QUEUES=1) any capture
        # that forks a stream segfaults in the ROCm 7.2 runtime (tools/capture_one_queue.py: plain torch ops): one stream then
        self.use_side_streams = os.environ.get("GPU_MAX_HW_QUEUES", "") != "1"
        self.bank_side_streams = True       # bank chains beside the Sinkhorn solve (head.head_forward)
        self._rng_state = None
        self._push_fn = None
        self._pushed = False
        # capture order of the loss-only step: (clustering launches, local-branch launches) per turn, last repeats
        self.capture_order = ((7, 5), (7, 1 << 30))
        self.bank_early = 2                 # bank chains started beside the clustering instead of the Sinkhorn (0..2)
        self.group_clustering = True        # text + video clustering in the same launches (no-grad forward)
        self.fuse_clustering = True
        # Training step: clustering forward on the grouped HIP kernels + the stage backward as grouped HIP kernels too
        # (cluster_fused.ClusterStagesFn -> cluster_backward_hip: nine launches per stage for both modalities) -- the default,
        # launched eagerly (5.3 ms against 11.8 with the autograd-traced torch ops) and replayed from a captured graph (3.6 against
        # 4.2) alike (tools/train_times.py, MI355X, round 3).  False: the autograd-traced torch ops of cluster.py on two side
        # streams, kept as the cross-check of the gradient tests.  (None is accepted as "default" for older callers.)
        self.fused_training_clustering = True
        self.shard_clustering = True               # with the sharded loss: every rank clusters its own samples only
        # Loss-only steps on W > 1 ranks, STEP-INTERLEAVED: every step, every rank takes part in the packed
        # all-gather and pushes the gathered batch into its replica of the memory bank (same ring, same prepared shadow, same noise
        # stream as a single-rank run); the step's LOSS is evaluated by ONE rank, the step's owner (step index mod W), with the
        # full replicated kernels.  Consecutive loss-only steps depend on each other through the bank alone, and the bank depends
        # on the inputs alone, so the ranks work on W consecutive steps at once: per W steps a rank pays one loss evaluation and
        # W exchange-and-push steps instead of W latency chains.  forward() returns None on the ranks that do not own the step.
        # Only without gradients (a training step needs every rank's gradient before the next step: the sharded loss is its form).
        self.interleave_steps = False
        self._step_index = 0
        # ... with the owner's loss BESIDE the following steps (OwnedSlot): the owner copies the bank's prepared shadow (19 MB at
        # configs[1], one bf16 pass on the bank side) before it pushes the batch like every other rank, and evaluates the loss
        # from that copy on the slot's own stream while this stream goes on with the exchange-and-push steps of the other owners.
        # Opt-in: the losses an owned step returns are then produced on that stream -- wait_owned_loss() before reading them.
        self.interleave_overlap = False
        # OwnedSlots in rotation, each with a loss stream of its own.  Measured (W = 8 / 4 emulated, us per round; serial 502 / 407):
        # one slot 436 / 421 (the next exchange half waits for this loss); two slots with ONE loss stream 491-503 / 423 (two loss
        # graphs queued on one stream hold the exchange graphs up); two slots, a stream each 362-371 / 349; three 432 / 344
        self.owned_slots = 2
        self._owned_ring = []
        self._owned_turn = 0
        self._owned = None                  # the slot of the last overlapped owned step
        self._pipeline = None               # a StepPipeline while consecutive steps are being captured overlapped (bench.py)
        self._ctm_cache = {}

    # ------------------------------------------------------------------ construction helpers
    @staticmethod
    def _scorer(width):
        return nn.Sequential(nn.Linear(width, 2 * width), nn.ReLU(inplace=True), nn.Linear(2 * width, 1))

    @staticmethod
    def _init_weights(module):
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=0.02)       # modeling.py:648-658
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    def _init_memory_bank(self):
        cpu = torch.device("cpu")
        # generation of the bank's STORAGE: bumped whenever the tensors a captured graph may have baked in (the five
        # bank tensors, the device ring head) are replaced -- main_retrieval.GraphedStep re-captures on a change
        self._mb_gen = getattr(self, "_mb_gen", 0) + 1
        # True: the step neither moves the ring head nor pushes (graph-capture warm-up must leave the bank alone)
        self.bank_frozen = getattr(self, "bank_frozen", False)
        self._mb = {
            "mb_ind": torch.tensor([], dtype=torch.long, device=cpu),
            "mb_feat_t": torch.empty((0, 0, 0), dtype=torch.float, device=cpu),
            "mb_feat_v": torch.empty((0, 0, 0), dtype=torch.float, device=cpu),
            "mb_mask_t": torch.empty((0, 0), dtype=torch.float, device=cpu),
            "mb_mask_v": torch.empty((0, 0), dtype=torch.float, device=cpu),
        }
        # ring head: a host integer for CPU banks; for GPU banks a device int32 (`_mb_head_dev`) that the step
        # prologue moves and nr_bank_ring_push reads, so that a captured HIP graph pushes to a new place at every
        # replay.  `_ring_advanced`: this step's prologue has already moved the head.
        self._mb_head = 0
        self._mb_head_dev = None
        self._ring_advanced = False
        self.mb_batch = 0
        # persistent prepared form of the bank (normalised * mask as bf16 hi/lo + norms), kept in step with the ring:
        # built lazily from the fp32 bank, extended at every push with the batch's own prepared rows, dropped whenever
        # the bank is touched from outside.  `_last_prepared`: the batch's prepared tokens of the running step.
        self._mb_shadow = None
        self._last_prepared = {}
        self.use_bank_shadow = True
        # similarity / bank work sharded over the ranks instead of the reference's replicated loss (loss-only step:
        # head.head_forward_sharded; training step: neighborretr_amd.sharded).  None = automatic: on from a gathered
        # batch of 512 (BASELINE configs[2]), off below -- at B = 128 the step is a latency chain that sharding cannot
        # shorten; True / False force it.  Its collectives run eagerly (not graph-captured).
        self.shard_loss = None

    # The bank attributes keep the reference's names and FIFO meaning (newest sample first; written
    # wholesale by MemoryBankManager, memory_bank.py:206-211).  Internally the bank is a RING: a push
    # rewrites only the batch rows (nr_bank_ring_push) instead of shifting the whole bank.  Reading an
    # attribute from outside materialises the FIFO order first; the forward pass reads the raw ring,
    # whose order is irrelevant (the bank is only consumed through means over its samples).
    def _bank_shadow(self, mb_feat_t=None, mb_feat_v=None):
        """(text, video) ops.Prepared of the bank for the loss-only step, or None (CPU bank, shadow disabled, grad
        mode: the training path keeps its own prepare launches, whose outputs the backward saves -- or the bank
        features handed to _compute_losses are not this model's own ring tensors)."""
        mb = self._mb
        if (mb_feat_t is not None and mb_feat_t is not mb["mb_feat_t"]) or (mb_feat_v is not None and mb_feat_v is not mb["mb_feat_v"]):
            return None
        if not self.use_bank_shadow or torch.is_grad_enabled() or mb["mb_feat_v"].numel() == 0 or not mb["mb_feat_v"].is_cuda:
            return None
        if self._mb_shadow is None:
            self._mb_shadow = (ops.prepare_tokens(mb["mb_feat_t"], mb["mb_mask_t"], want_lo=True),
                               ops.prepare_tokens(mb["mb_feat_v"], mb["mb_mask_v"], want_lo=True))
        return self._mb_shadow

    def _bank_fifo(self):
        if self._mb_head_dev is not None or self._mb_head:
            self._mb_shadow = None                 # the roll below re-orders the rows: rebuild on next use
        if self._mb_head_dev is not None:
            self._mb_head = int(self._mb_head_dev.item())          # (a sync: only when the bank is read from outside)
            self._mb_head_dev = None
            self._mb_gen += 1                                      # a captured graph holds the old head tensor
        if self._mb_head:
            h = self._mb_head
            self._mb = {k: torch.roll(v, shifts=-h, dims=0) for k, v in self._mb.items()}
            self._mb_head = 0
            self._mb_gen += 1

    def _ring_ready(self, b, whole=False):
        """(head tensor, advance, capacity) for the step prologue when the coming push takes the ring path, else None.
        whole: a batch that replaces the bank (b >= capacity) counts too -- nr_bank_absorb_gathered takes it, at head 0."""
        mb = self._mb
        cap = mb["mb_feat_v"].size(0)
        if cap == 0 or (b >= cap and not whole) or not all(t.is_cuda and t.is_contiguous() for t in mb.values()):
            return None
        if self._mb_head_dev is None or self._mb_head_dev.device != mb["mb_feat_v"].device:
            self._mb_head_dev = torch.tensor([self._mb_head], dtype=torch.int32, device=mb["mb_feat_v"].device)
            self._mb_head = 0
            self._mb_gen += 1
        return self._mb_head_dev, b, cap

    def _bank_get(self, name):
        self._bank_fifo()
        return self._mb[name]

    def _bank_set(self, name, value):
        self._bank_fifo()
        if name.startswith("mb_mask") and torch.is_tensor(value) and value.dtype != torch.float32:
            value = value.float()        # masks live as fp32 (the reference's own initial dtype, :182-183)
        self._mb[name] = value
        self._mb_shadow = None
        self._mb_gen += 1

    mb_ind = property(lambda self: self._bank_get("mb_ind"), lambda self, v: self._bank_set("mb_ind", v))
    mb_feat_t = property(lambda self: self._bank_get("mb_feat_t"), lambda self, v: self._bank_set("mb_feat_t", v))
    mb_feat_v = property(lambda self: self._bank_get("mb_feat_v"), lambda self, v: self._bank_set("mb_feat_v", v))
    mb_mask_t = property(lambda self: self._bank_get("mb_mask_t"), lambda self, v: self._bank_set("mb_mask_t", v))
    mb_mask_v = property(lambda self: self._bank_get("mb_mask_v"), lambda self, v: self._bank_set("mb_mask_v", v))

    # ------------------------------------------------------------------ scorer weights (bf16 split)
    def _prec(self, for_head=True):
        if self.precision == "bf16x3":
            return hip.PREC_BF16X3
        if self.precision == "bf16_all":
            return hip.PREC_BF16
        return head.PREC_MIXED if for_head else hip.PREC_BF16X3

    def scorer_weights(self, name):
        """bf16 hi/lo images of one scorer MLP, re-split when the parameters were updated."""
        mlp = getattr(self, name)
        ps = (mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias)          # Sequential(Linear, ReLU, Linear)
        ver = tuple(p._version for p in ps) + tuple(p.data_ptr() for p in ps)
        hit = self._scorer_cache.get(name)
        if hit is None or hit[0] != ver:
            sw = head.ScorerWeights(mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias)
            self._scorer_cache[name] = (ver, sw)
            return sw
        return hit[1]

    def _global_scorers(self, text_feat, video_feat):
        """sw_t1 / sw_v1 keyword arguments of the head: the *_weight_fc1 scorers, needed by global_level only when a
        sample keeps more than one global token (modeling.py:518-523; one token => softmax weight 1)."""
        (_, t1), (_, v1) = self._token_counts(text_feat.shape[1], video_feat.shape[1])
        if t1 == 1 and v1 == 1:
            return {}
        return dict(sw_t1=self.scorer_weights("text_weight_fc1"), sw_v1=self.scorer_weights("video_weight_fc1"))

    # ------------------------------------------------------------------ memory bank (modeling.py:222-249)
    def update_memory_bank(self, idx, text_feat, video_feat, text_mask, video_mask):
        new = {"mb_ind": idx, "mb_feat_v": video_feat.detach(), "mb_feat_t": text_feat.detach(),
               "mb_mask_t": text_mask, "mb_mask_v": video_mask}
        mb = self._mb
        if mb["mb_feat_v"].size(0) == 0:                       # empty bank adopts the batch (:224-231)
            self._mb = {k: (v.float() if k.startswith("mb_mask") else v.clone()) for k, v in new.items()}
            self._mb_shadow = None
            self._mb_head, self._mb_head_dev = 0, None
            self._mb_gen += 1
            self.mb_batch = idx.size(0)
            return
        cap, b = mb["mb_feat_v"].size(0), idx.size(0)
        on_gpu = all(t.is_cuda and t.is_contiguous() for t in mb.values())
        if b >= cap or not on_gpu:
            # B >= capacity: the bank becomes the first rows of the batch (:244-249); CPU banks: plain cat
            if b >= cap:                           # (cat(batch, bank)[:cap] without the cat: the batch's first `cap` rows)
                rows = {k: new[k][:cap] for k in self._mb}
                if on_gpu and all(rows[k].is_cuda and rows[k].is_contiguous() and rows[k].dtype == v.dtype and rows[k].shape == v.shape
                                  for k, v in self._mb.items()):
                    # in place, one launch: the storage, the device ring head (now 0; no read-back -- this runs inside captured
                    # steps) and the prepared shadow stay, the shadow taking the batch's own prepared rows as on the ring path
                    if self._mb_head_dev is not None:
                        self._mb_head_dev.zero_()
                    self._mb_head = 0
                    dsts, srcs = list(self._mb.values()), [rows[k] for k in self._mb]
                    sh, lp = self._mb_shadow, self._last_prepared
                    if sh is not None and lp.get("pt") is not None and lp["pt"].lo is not None and lp["pv"].lo is not None \
                            and sh[0].lo is not None and sh[1].lo is not None:
                        for prep_b, prep_n, N in ((sh[0], lp["pt"], text_feat.shape[1]), (sh[1], lp["pv"], video_feat.shape[1])):
                            n = cap * N
                            dsts += [prep_b.hi, prep_b.lo, prep_b.norm]
                            srcs += [prep_n.hi.view(-1, prep_b.d)[:n], prep_n.lo.view(-1, prep_b.d)[:n], prep_n.norm.view(-1)[:n]]
                        dsts = [t.view(-1) for t in dsts]
                        srcs = [t.reshape(-1) for t in srcs]
                    else:
                        self._mb_shadow = None
                    self._last_prepared = {}
                    ops.copy_group(dsts, srcs)
                    return
                self._bank_fifo()
                self._mb = {k: rows[k].to(v.dtype, copy=True).contiguous() for k, v in self._mb.items()}
            else:
                self._bank_fifo()
                self._mb = {k: torch.cat((new[k].to(v.dtype), v), 0)[:cap].contiguous() for k, v in self._mb.items()}
            self._mb_shadow = None
            self._mb_gen += 1
            return
        ring = self._ring_ready(b)
        if not self._ring_advanced:                            # called outside loss_step: move the head here
            ring[0].sub_(b).remainder_(cap)
        self._ring_advanced = False
        names = list(mb)
        banks, rows = [mb[k] for k in names], [new[k].to(mb[k].dtype) for k in names]
        sh, lp = self._mb_shadow, self._last_prepared
        if sh is not None:
            if lp.get("pt") is not None and lp["pt"].lo is not None and lp["pv"].lo is not None:
                # the batch's prepared rows (computed by this step's local branch) extend the shadow in the same launch
                Nt, Nv = text_feat.shape[1], video_feat.shape[1]
                for prep_b, prep_n, N in ((sh[0], lp["pt"], Nt), (sh[1], lp["pv"], Nv)):
                    d = prep_b.d
                    banks += [prep_b.hi.view(cap, N * d), prep_b.lo.view(cap, N * d), prep_b.norm.view(cap, N)]
                    rows += [prep_n.hi.view(b, N * d), prep_n.lo.view(b, N * d), prep_n.norm.view(b, N)]
            else:
                self._mb_shadow = None             # pushed without prepared rows (direct call): rebuild on next use
        self._last_prepared = {}
        pipe = self._pipeline
        ops.bank_ring_push(banks, rows, 0, head_dev=pipe.head if (pipe is not None and pipe.head is not None) else ring[0])

    # ------------------------------------------------------------------ forward (modeling.py:251-312)
    def forward(self, text_ids, text_mask, video, video_mask=None, idx=None, global_step=0, logger=None):
        text_mask = text_mask.view(-1, text_mask.shape[-1])
        video_mask = video_mask.view(-1, video_mask.shape[-1])
        if not self.feature_mode:                      # modeling.py:253-265: token ids [b, Nt], frames [b*Nv, 3, H, W]
            text_ids = text_ids.view(-1, text_ids.shape[-1])
            video = torch.as_tensor(video)
            video = video.reshape((-1,) + tuple(video.shape[-3:]))
        text_feat, video_feat = self.get_text_video_feat(text_ids, text_mask, video, video_mask, shaped=True)
        if not self.training:
            return None
        world = int(getattr(self.config, "world_size", 1))
        if world > 1:
            from . import comm
            comm.begin_step()
            # the reference's 5 all_gathers + barrier (modeling.py:274-280) as one packed collective.  Its backward depends
            # on how the loss is evaluated (replicated: slice; sharded: reduce-scatter), decided before the gather
            self.config.shard_loss = self._shard_now(world, text_feat, gathered_rows=text_feat.shape[0] * world,
                                                     video_tokens=video_feat.shape[1])
            from .dist import packed_allgather, packed_gather_raw, unpack_raw
            if self._interleaving():
                k, self._step_index = self._step_index, self._step_index + 1
                owner = k % world == comm.get_rank()
                if text_feat.is_cuda:
                    with torch.no_grad():
                        recv, lay = packed_gather_raw(text_feat, video_feat, idx, text_mask, video_mask, self.config)
                        if not owner and self._absorb_gathered(recv, lay):
                            return None                        # the whole step in three launches: pack, all-gather, absorb
                        if owner and self.interleave_overlap:
                            slot = self._owned_prepare(recv, lay)
                            if slot is not None:               # the batch is in the bank already; the loss runs beside what follows
                                return self._owned_loss_beside(slot)
                        text_feat, video_feat, idx, text_mask, video_mask = unpack_raw(recv, lay)
                else:
                    text_feat, video_feat, idx, text_mask, video_mask = packed_allgather(
                        text_feat, video_feat, idx, text_mask, video_mask, self.config)
                if not owner:
                    self.bank_only_step(text_feat, video_feat, text_mask, video_mask, idx)
                    return None
                return self.loss_step(text_feat, video_feat, text_mask, video_mask, idx)
            text_feat, video_feat, idx, text_mask, video_mask = packed_allgather(
                text_feat, video_feat, idx, text_mask, video_mask, self.config)
        return self.loss_step(text_feat, video_feat, text_mask, video_mask, idx)

    def _interleaving(self):
        return bool(self.interleave_steps) and not torch.is_grad_enabled() and int(getattr(self.config, "world_size", 1)) > 1

    def _absorb_ready(self, lay):
        """(ring, shadow) when a gathered batch of this layout can go into the bank through nr_bank_absorb_gathered -- the bank is
        a device ring of fp32 tensors of the batch's token shapes -- else None.  Does not look at
        `bank_frozen`."""
        (Nt, d), (Nv, _) = lay["shapes"][0], lay["shapes"][1]
        B = lay["W"] * lay["b"]
        if d % 256 or d > 1024 or Nt > 64 or Nv > 64:
            return None
        mb = self._mb
        if tuple(mb["mb_feat_t"].shape[1:]) != (Nt, d) or tuple(mb["mb_feat_v"].shape[1:]) != (Nv, d) or mb["mb_ind"].dtype != torch.int64:
            return None
        if any(mb[k].dtype != torch.float32 for k in ("mb_feat_t", "mb_feat_v", "mb_mask_t", "mb_mask_v")):
            return None
        ring = self._ring_ready(B, whole=True)
        if ring is None:
            return None
        shadow = self._bank_shadow()                       # (built before the first push, like the owner's step does)
        if shadow is not None and (shadow[0].lo is None or shadow[1].lo is None):
            return None
        return ring, shadow

    def _absorb_gathered(self, recv, lay):
        """A step this rank does not own, from the exchange step's receive buffer in ONE launch (nr_bank_absorb_gathered): ring
        head, noise counter, fp32 bank rows and the prepared shadow rows.  False when the bank is not a device ring of the
        batch's shapes, or frozen (the caller then unpacks and takes bank_only_step)."""
        pre = None if self.bank_frozen else self._absorb_ready(lay)
        if pre is None:
            return False
        ring, shadow = pre
        ops.bank_absorb_gathered(recv, lay, self._mb, shadow, ring[0], ring[2], self._rng_state_on(recv.device))
        self._last_prepared = {}
        return True

    # ------------------------------------------------------------------ overlapped owned step (interleave_overlap)
    def _owned_prepare(self, recv, lay, slot_index=None):
        """Exchange half of an overlapped owned step, behind the all-gather, on the current stream: the gathered batch unpacked
        into a slot, the bank as the loss must see it copied into the slot (OwnedSlot.take), then the batch pushed into the
        live bank exactly as the ranks that do not own the step push it.  The slots take turns (`owned_slots`; slot_index
        names one: a captured graph is tied to its slot).  -> the slot, or None when the bank cannot absorb (the caller then
        runs the serial owned step)."""
        pre = self._absorb_ready(lay)
        if pre is None or pre[1] is None:
            return None
        ring, shadow = pre
        dev = recv.device
        rng = self._rng_state_on(dev)
        masks = (self._mb["mb_mask_t"], self._mb["mb_mask_v"])
        copy_lo = head.precision_plan(self._prec())[2] == hip.PREC_BF16X3
        capturing = torch.cuda.is_current_stream_capturing()
        slots = self._owned_ring
        if (len(slots) != max(1, int(self.owned_slots)) or slots[0].rng.device != dev
                or not all(s_.fits(lay, shadow, masks, copy_lo) for s_ in slots)):
            if capturing:
                raise RuntimeError("overlapped owned step: run one eager step before the capture (the slots' buffers must not live "
                                   "in a graph's memory pool)")
            slots = self._owned_ring = [OwnedSlot(lay, shadow, masks, rng, copy_lo) for _ in range(max(1, int(self.owned_slots)))]
        if slot_index is None:
            slot_index, self._owned_turn = self._owned_turn % len(slots), self._owned_turn + 1
        slot = self._owned = slots[slot_index % len(slots)]
        if slot.loss_done is not None and not capturing:
            torch.cuda.current_stream().wait_event(slot.loss_done)     # the previous loss on this slot has read it
        ops.unpack_gathered(recv, lay["W"], lay["record"], lay["sizes"], lay["offs"], slot.batch, [False, False, False, True, True])
        slot.take(shadow, masks, rng)
        if self.bank_frozen:
            tf, vf, ix, tm, vm = slot.batch
            self.bank_only_step(tf, vf, tm, vm, ix)        # frozen: only the noise counter moves, as in every other form
        else:
            ops.bank_absorb_gathered(recv, lay, self._mb, shadow, ring[0], ring[2], rng)
            self._last_prepared = {}
        return slot

    def owned_exchange(self, text_feat, text_mask, video_feat, video_mask, idx, slot_index=None):
        """The exchange half of an overlapped owned step as a call of its own (what bench.py captures as one graph or as
        segmented graphs): packed all-gather -> _owned_prepare.  -> the slot; raises when the bank cannot absorb."""
        from . import comm
        from .dist import packed_gather_raw
        comm.begin_step()
        self._step_index += 1
        with torch.no_grad():
            recv, lay = packed_gather_raw(text_feat, video_feat, idx, text_mask.view(-1, text_mask.shape[-1]),
                                          video_mask.view(-1, video_mask.shape[-1]), self.config)
            slot = self._owned_prepare(recv, lay, slot_index)
        if slot is None:
            raise RuntimeError("overlapped owned step: the memory bank cannot absorb a gathered batch (not a device ring of fp32 "
                               "tensors of the batch's token shapes, or no prepared shadow)")
        return slot

    def owned_loss(self, slot=None):
        """The loss half: loss_step on the slot's batch against the slot's copy of the bank -- bank frozen (the batch has been
        pushed by the exchange half), the noise drawn from the slot's copy of the counter -- on the current stream."""
        slot = self._owned if slot is None else slot
        live = self._mb
        keep = (self._mb, self._mb_shadow, self._rng_state, self.bank_frozen, self._last_prepared)
        self._mb = {"mb_ind": live["mb_ind"], "mb_feat_t": live["mb_feat_t"], "mb_feat_v": live["mb_feat_v"],     # (shapes only)
                    "mb_mask_t": slot.masks[0], "mb_mask_v": slot.masks[1]}
        self._mb_shadow, self._rng_state, self.bank_frozen, self._last_prepared = slot.shadow, slot.rng, True, slot.scratch
        try:
            tf, vf, ix, tm, vm = slot.batch
            with torch.no_grad():
                slot.losses = self.loss_step(tf, vf, tm, vm, ix)
        finally:
            self._mb, self._mb_shadow, self._rng_state, self.bank_frozen, self._last_prepared = keep
            slot.scratch.clear()
        return slot.losses

    def _owned_loss_beside(self, slot):
        cur = torch.cuda.current_stream()
        if torch.cuda.is_current_stream_capturing():
            return self.owned_loss(slot)               # one capture for the whole step: the halves stay in a row
        side = slot.stream
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            losses = self.owned_loss(slot)
            slot.loss_done = torch.cuda.Event()
            slot.loss_done.record(side)
        return losses

    def wait_owned_loss(self):
        """Orders the current stream behind the loss halves of the overlapped owned steps so far (no host synchronisation)."""
        for slot in self._owned_ring:              # (the slots' loss streams run side by side: every one of them)
            if slot.loss_done is not None:
                torch.cuda.current_stream().wait_event(slot.loss_done)

    def bank_only_step(self, text_feat, video_feat, text_mask, video_mask, idx):
        """A step this rank does not own (interleave_steps): everything of loss_step that outlives the step -- the ring head and
        the noise stream move exactly as in loss_step (same prologue launch), the batch's prepared rows extend the bank's
        shadow, the batch takes the oldest rows' place (modeling.py:309-310) -- and none of the loss.  Three launches."""
        raw_scale = self.clip.logit_scale
        sizes = self._noise_sizes(text_feat.shape[1], video_feat.shape[1])
        B = text_feat.shape[0]
        with torch.no_grad():
            shadow = self._bank_shadow()                                  # built before the first push, like the owner's step does
            ring = None if self.bank_frozen else self._ring_ready(B)
            text_mask, video_mask, _, _ = ops.step_prologue(text_mask, video_mask, raw_scale, self._rng_state_on(text_feat.device),
                                                            B * sum(sizes.values()), ring=ring)
            self._ring_advanced = ring is not None
            if self.bank_frozen:
                return
            if shadow is not None:
                pt, pv = ops.prepare_tokens_pair(text_feat, text_mask, video_feat, video_mask, want_lo=True, want_colsum=True)
                self._last_prepared = {"pt": pt, "pv": pv}
            self.update_memory_bank(idx, text_feat, video_feat, text_mask, video_mask)

    def loss_step(self, text_feat, video_feat, text_mask, video_mask, idx):
        """Everything after the exchange step: the five losses on the (gathered) global batch and the
        memory-bank push (modeling.py:283-312).  Collective-free, so it captures into a HIP graph."""
        noise = None
        raw_scale = self.clip.logit_scale
        self._raw_masks = (text_mask, video_mask)      # as the caller handed them over (the early-forked local branch converts its own copy)
        if text_feat.is_cuda:
            # masks as fp32 multipliers, exp(logit_scale) and the DPC-KNN tie-break noise in ONE launch
            scale_in_kernel = not (torch.is_grad_enabled() and raw_scale.requires_grad)
            sizes = self._noise_sizes(text_feat.shape[1], video_feat.shape[1])
            B = text_feat.shape[0]
            ring = None if self.bank_frozen else self._ring_ready(B)
            text_mask, video_mask, ls_exp, flat = ops.step_prologue(
                text_mask, video_mask, raw_scale if scale_in_kernel else None,
                self._rng_state_on(text_feat.device), B * sum(sizes.values()), ring=ring)
            self._ring_advanced = ring is not None
            if self._pipeline is not None:
                from .capture_guard import record_event as _rec
                self._pipeline.prologue_done = _rec(torch.cuda.current_stream())
            if self._pipeline is not None and self._pipeline.decoupled and ring is not None:
                # this step's own copy of the ring head, in a PERSISTENT word per slot: the push reads it late, on its own stream --
                # a per-step allocation of this stream would be handed out again to the next step before the push has run
                # (measured: "Memory access fault", the push wrote at a recycled word's value)
                self._pipeline.head = self._head_slot(self._pipeline.slot, ring[0])
                self._pipeline.head.copy_(ring[0])
            logit_scale = ls_exp.reshape(()) if scale_in_kernel else raw_scale.exp()
            noise = self._slice_noise(flat, B, sizes)
        else:
            text_mask, video_mask = text_mask.float(), video_mask.float()
            logit_scale = raw_scale.exp()
        cfg = self.config
        # the push (modeling.py:309-310): the loss-only head may run it itself as soon as the bank has been read
        # (head.head_forward `bank_push`); otherwise it follows the losses here
        self._pushed = False

        def push():
            self.update_memory_bank(idx, text_feat, video_feat, text_mask, video_mask)
            self._pushed = True
            if self._pipeline is not None:
                from .capture_guard import record_event
                self._pipeline.push_done = record_event(torch.cuda.current_stream())
        self._push_fn = None if self.bank_frozen else push
        losses = self._compute_losses(text_feat, video_feat, text_mask, video_mask,
                                      self._mb["mb_feat_t"], self._mb["mb_feat_v"], self._mb["mb_mask_t"], self._mb["mb_mask_v"],
                                      cfg.centrality_scale, cfg.beta, cfg.num_neighbors, cfg.temperature,
                                      logit_scale, noise=noise)
        self._push_fn = None
        self._raw_masks = None
        if not self.bank_frozen and not self._pushed:
            with torch.no_grad():
                push()
        if self._pipeline is not None:
            self._pipeline.own(text_mask, video_mask, logit_scale, noise, losses, self._last_prepared)
        return losses

    # ------------------------------------------------------------------ losses (modeling.py:314-360)
    def _hp(self, centrality_scale, beta, num_neighbors, temperature):
        c = self.config
        return dict(centrality_scale=centrality_scale, beta=beta, num_neighbors=num_neighbors,
                    temperature=temperature, uniform_weight=c.uniform_weight, neighbor_weight=c.neighbor_weight,
                    kl_weight=c.kl_weight, centrality_multi_token=getattr(c, "centrality_multi_token", "raise"))

    def _compute_losses(self, text_feat, video_feat, text_mask, video_mask,
                        mb_feat_t, mb_feat_v, mb_mask_t, mb_mask_v,
                        centrality_scale, beta, num_neighbors, temperature, logit_scale, noise=None):
        hp = self._hp(centrality_scale, beta, num_neighbors, temperature)
        from .functional import head_losses
        if noise is None and text_feat.is_cuda:
            noise = self._draw_noise(text_feat.shape[0], text_feat.shape[1], video_feat.shape[1], text_feat.device)
        mods = tuple(getattr(self, f"{w}_{k}") for w in ("text", "video") for k in ("ctm0", "block0", "ctm1", "block1"))
        nz = noise or {}
        world = int(getattr(self.config, "world_size", 1))
        shard_now = self._shard_now(world, text_feat, video_tokens=video_feat.shape[1])
        if shard_now and torch.is_grad_enabled():
            # training step with the loss sharded over the ranks (neighborretr_amd.sharded)
            from . import comm
            from .sharded import sharded_training_losses
            losses = sharded_training_losses(self, text_feat, video_feat, text_mask, video_mask, mb_feat_t, mb_feat_v,
                                             mb_mask_t, mb_mask_v, hp, logit_scale, comm.get_rank(), world, noise)
            return losses[0], losses[1], losses[2], losses[3], losses[4]
        if (shard_now and not torch.is_grad_enabled()
                and self._can_fuse_clustering(text_feat, mods) and self._can_fuse_clustering(video_feat, mods)):
            from . import comm
            # The clustering is sharded by samples too: every rank clusters ITS b samples and the [b, c, d] global tokens are
            # all-gathered.  Its masked stage fills distances with the maximum over the WHOLE gathered batch
            # (cluster.py:473-475, `dist_matrix.max()`), so the ranks exchange that maximum (one all-reduce of two floats)
            # between the stage's front and back kernels -- without it the densities of samples with fewer than k valid
            # tokens would differ from the replicated result.
            rank_ = comm.get_rank()
            if self.shard_clustering:
                def join():
                    return self._gather_global(*self._merge_sharded(text_feat, video_feat, text_mask, video_mask, nz, rank_, world), world)
            else:
                def join():
                    return self._merge_grouped(text_feat, video_feat, text_mask, video_mask, nz)
            # the clustering (+ the gather of the global tokens) runs on THIS stream, the local branch beside it
            losses = head.head_forward_sharded(text_feat, video_feat, text_mask, video_mask, mb_feat_t, mb_feat_v,
                                               mb_mask_t, mb_mask_v, None, None, self.scorer_weights("text_weight_fc"),
                                               self.scorer_weights("video_weight_fc"), hp, logit_scale, self._prec(),
                                               rank_, world, bank_prepared=self._bank_shadow(mb_feat_t, mb_feat_v),
                                               prepared_out=self._last_prepared, join=join,
                                               local_stream=self._local_stream(text_feat.device),
                                               **self._global_scorers(text_feat, video_feat))
            return losses[0], losses[1], losses[2], losses[3], losses[4]
        if (text_feat.is_cuda and self.use_side_streams and self.group_clustering and not torch.is_grad_enabled()
                and self._can_fuse_clustering(text_feat, mods) and self._can_fuse_clustering(video_feat, mods)):
            # loss-only step: the text and video clustering advance together inside grouped launches on THIS
            # stream (the critical path); the local branch runs beside them on a side stream, the two bank
            # chains beside the Sinkhorn solve (head.head_forward).
            gt = gv = None

            self._join_global = self._merge_grouped_steps(text_feat, video_feat, text_mask, video_mask, nz)
        elif (text_feat.is_cuda and self.fuse_clustering and torch.is_grad_enabled()
              and (self.fused_training_clustering is None or self.fused_training_clustering)
              and text_feat.shape[1] <= 64 and video_feat.shape[1] <= 64 and text_feat.shape[2] % 128 == 0):
            # training step: the clustering forward on the grouped HIP kernels, the backward hand-derived from what they
            # leave in their workspaces (cluster_fused.ClusterStagesFn, cluster_backward.stage_backward)
            from .cluster_fused import build_stage_weights, cluster_stages_train, route_on_this_stream, stage_params
            mods0 = ((self.text_ctm0, self.text_block0), (self.video_ctm0, self.video_block0))
            mods1 = ((self.text_ctm1, self.text_block1), (self.video_ctm1, self.video_block1))
            side = self._cluster_stream(text_feat.device)
            routed, (tf_c, vf_c) = ({}, (text_feat, video_feat))
            if side is not None:
                # the clustering runs on its own stream: its parameters and the leaf features enter it through one identity
                # node of THIS stream, so that their gradients are accumulated on this stream (cluster_fused.route_on_this_stream)
                routed, (tf_c, vf_c) = route_on_this_stream([p for ctm, blk in mods0 + mods1 for p in stage_params(ctm, blk)],
                                                            (text_feat, video_feat))

            def stages():
                # the bf16 pairs of all four stages' weights (stale after every optimizer step) in ONE split launch
                build_stage_weights(self._ctm_cache, [("text0", self.text_ctm0, self.text_block0), ("video0", self.video_ctm0, self.video_block0),
                                                      ("text1", self.text_ctm1, self.text_block1), ("video1", self.video_ctm1, self.video_block1)])
                t, v = cluster_stages_train(mods0, self._ctm_cache, ("text0", "video0"), tf_c, text_mask, nz.get("t0"),
                                            vf_c, video_mask, nz.get("v0"), routed=routed)
                return cluster_stages_train(mods1, self._ctm_cache, ("text1", "video1"), t, None, nz.get("t1"), v, None, nz.get("v1"),
                                            routed=routed)
            from . import backward as _bw
            if (side is not None and self.interleave_training_forward and _bw.SPLIT_HEAD_NODES
                    and self._local_stream(text_feat.device) is not None):
                # The training forward on the loss-only step's schedule: the clustering's launches are set up here (nothing is
                # issued yet) and driven one by one by the head, interleaved in capture order with its local branch and the
                # bank chains (459 -> ~340 us for the forward of the captured step: a branch captured behind all of the
                # clustering started ~200 us late).  The clustering's autograd nodes are created afterwards, on the stream
                # their backward is to run on; their forwards only wrap the stages' outputs and saved intermediates.
                from .cluster_fused import ctm_stage_group
                build_stage_weights(self._ctm_cache, [("text0", self.text_ctm0, self.text_block0), ("video0", self.video_ctm0, self.video_block0),
                                                      ("text1", self.text_ctm1, self.text_block1), ("video1", self.video_ctm1, self.video_block1)])
                o0, g0, sv0 = ctm_stage_group([("text0", text_feat, text_mask, *mods0[0], nz.get("t0")),
                                               ("video0", video_feat, video_mask, *mods0[1], nz.get("v0"))], self._ctm_cache,
                                              stepwise=True, want_saved=True)
                o1, g1, sv1 = ctm_stage_group([("text1", o0[0], None, *mods1[0], nz.get("t1")),
                                               ("video1", o0[1], None, *mods1[1], nz.get("v1"))], self._ctm_cache,
                                              stepwise=True, want_saved=True)

                def launches():
                    yield from g0
                    yield from g1
                    return o1[0], o1[1]

                def make_nodes():
                    with torch.cuda.stream(side):
                        t, v = cluster_stages_train(mods0, self._ctm_cache, ("text0", "video0"), tf_c, text_mask, nz.get("t0"),
                                                    vf_c, video_mask, nz.get("v0"), pre=((o0[0], o0[1]), sv0), routed=routed)
                        return cluster_stages_train(mods1, self._ctm_cache, ("text1", "video1"), t, None, nz.get("t1"), v, None,
                                                    nz.get("v1"), pre=((o1[0], o1[1]), sv1), routed=routed)
                losses = head_losses(self, text_feat, video_feat, text_mask, video_mask, mb_feat_t, mb_feat_v, mb_mask_t, mb_mask_v,
                                     None, None, hp, logit_scale, cluster=(launches(), make_nodes))
                return losses[0], losses[1], losses[2], losses[3], losses[4]
            if side is None:
                gt, gv = stages()
            else:
                # The clustering is issued on its own stream: in the forward it runs beside the head's local branch (joined
                # right before the global logits), and -- what matters more -- autograd runs its BACKWARD on that stream too,
                # beside the backward of the head's token side (backward.HeadLocalFn), once the short HeadGlobalFn node has
                # produced the gradients of the global tokens.  Fork from / join into the step's stream only.
                cur = torch.cuda.current_stream()
                wait_stream(side, cur)
                with torch.cuda.stream(side):
                    gt, gv = stages()
                gt.record_stream(cur)
                gv.record_stream(cur)

                def join():
                    wait_stream(cur, side)
                self._join_global = join
        elif text_feat.is_cuda and self.use_side_streams:
            # three independent branches: text clustering | video clustering | local products.
            # The two clustering branches run on side streams (in a captured HIP graph: parallel
            # branches) and are joined right before the global logits need them.
            cur = torch.cuda.current_stream()
            s_t, s_v = self._side_streams(text_feat.device)
            tf_c, vf_c = text_feat, video_feat
            if torch.is_grad_enabled():
                # leaf features enter the side streams through an identity node of THIS stream: the head's local branch uses
                # them on this stream as well, and their AccumulateGrad nodes then see one stream (cluster_fused.route_on_this_stream)
                from .cluster_fused import route_on_this_stream
                _, (tf_c, vf_c) = route_on_this_stream([], (text_feat, video_feat))
            wait_stream(s_t, cur)
            wait_stream(s_v, cur)
            with torch.cuda.stream(s_t):
                gt = self._merge_one("text", tf_c, text_mask, nz.get("t0"), nz.get("t1"))
            with torch.cuda.stream(s_v):
                gv = self._merge_one("video", vf_c, video_mask, nz.get("v0"), nz.get("v1"))
            gt.record_stream(cur)
            gv.record_stream(cur)

            def join():
                wait_stream(cur, s_t)
                wait_stream(cur, s_v)
            self._join_global = join
        else:
            gt, gv = self.merge_global_features(text_feat, video_feat, text_mask, video_mask, noise)
        losses = head_losses(self, text_feat, video_feat, text_mask, video_mask, mb_feat_t, mb_feat_v,
                             mb_mask_t, mb_mask_v, gt, gv, hp, logit_scale)
        self._take_join()      # joined by now; never leave a stale closure behind
        return losses[0], losses[1], losses[2], losses[3], losses[4]

    @contextlib.contextmanager
    def graph_capture_mode(self):
        """For the warm-up AND the capture of a training step as a HIP graph: the warm-up must run exactly the kernels -- and
        initialise exactly the library handles -- the capture will record (hipBLASLt refuses to set itself up inside a capture).
        Every choice of the training step is now independent of the launch mode, so this only pins `fused_training_clustering`
        against a change between warm-up and capture."""
        old = self.fused_training_clustering
        self.fused_training_clustering = True if old is None else old
        try:
            yield self
        finally:
            self.fused_training_clustering = old

    def _shard_now(self, world, text_feat, gathered_rows=None, video_tokens=None):
        """Sharded loss: when asked for (`shard_loss` = True, also config.shard_loss for the exchange step's backward), or
        by default once the gathered batch reaches 512 samples (BASELINE configs[2]: the replicated loss would repeat
        677 GF on every rank) -- `shard_loss` = False keeps the reference's replicated form at any size.
        ONE decision for the exchange step (forward(): called before the gather with `gathered_rows`) and for the loss
        (_compute_losses: called on the gathered batch): the two must agree, the exchange step's backward depends on it.
        The AUTOMATIC choice also needs one global token per sample in training: sharded_training_losses covers that case
        only, the replicated head trains at ActivityNet token counts too (centrality_multi_token = "mean")."""
        if world < 2 or not text_feat.is_cuda or self._interleaving():
            return False
        rows = text_feat.shape[0] if gathered_rows is None else gathered_rows
        if self.shard_loss is not None:
            want = self.shard_loss
        else:
            want = rows >= 512
            if want and torch.is_grad_enabled() and video_tokens is not None:
                (_, t1), (_, v1) = self._token_counts(text_feat.shape[1], video_tokens)
                if (t1, v1) != (1, 1):
                    if not getattr(self, "_warned_multi_token_shard", False):
                        self._warned_multi_token_shard = True
                        import warnings
                        warnings.warn("gathered batch >= 512 with several global tokens per sample: the sharded training loss "
                                      "covers one global token; keeping the replicated loss")
                    want = False
        return bool(want and rows % world == 0)

    # ------------------------------------------------------------------ token clustering (modeling.py:446-481)
    def merge_global_features(self, text_feat, video_feat, text_mask, video_mask, noise=None):
        nz = noise or {}
        return (self._merge_one("text", text_feat, text_mask, nz.get("t0"), nz.get("t1")),
                self._merge_one("video", video_feat, video_mask, nz.get("v0"), nz.get("v1")))

    def _merge_grouped_steps(self, text_feat, video_feat, text_mask, video_mask, nz):
        """The two grouped clustering stages as a generator that issues ONE kernel launch per next() (on the
        then-current stream) and returns (gt, gv): head.head_forward interleaves these launches with the local
        branch's, because a captured HIP graph starts its nodes in capture order."""
        from .cluster_fused import ctm_stage_group
        (t, v), steps = ctm_stage_group(
            [("text0", text_feat, text_mask, self.text_ctm0, self.text_block0, nz.get("t0")),
             ("video0", video_feat, video_mask, self.video_ctm0, self.video_block0, nz.get("v0"))], self._ctm_cache, stepwise=True)
        yield from steps
        (t, v), steps = ctm_stage_group(
            [("text1", t, None, self.text_ctm1, self.text_block1, nz.get("t1")),
             ("video1", v, None, self.video_ctm1, self.video_block1, nz.get("v1"))], self._ctm_cache, stepwise=True)
        yield from steps
        return t, v

    def _merge_grouped(self, text_feat, video_feat, text_mask, video_mask, nz):
        """Both modalities through the two clustering stages in grouped launches (no-grad forward)."""
        from .cluster_fused import ctm_stage_group
        t, v = ctm_stage_group([("text0", text_feat, text_mask, self.text_ctm0, self.text_block0, nz.get("t0")),
                                ("video0", video_feat, video_mask, self.video_ctm0, self.video_block0, nz.get("v0"))],
                               self._ctm_cache)
        t, v = ctm_stage_group([("text1", t, None, self.text_ctm1, self.text_block1, nz.get("t1")),
                                ("video1", v, None, self.video_ctm1, self.video_block1, nz.get("v1"))],
                               self._ctm_cache)
        return t, v

    def _merge_sharded(self, text_feat, video_feat, text_mask, video_mask, nz, rank, world):
        """Both clustering stages on THIS rank's samples of the gathered batch (rows [rank b, (rank+1) b)) -> the rank's
        global tokens (gt [b,c,d], gv [b,c,d]); differentiable when gradients are enabled (ClusterStagesFn).  The noise rows
        are the rank's rows of the batch-wide draw, so the result equals the rank's rows of the replicated clustering."""
        from . import comm
        from .cluster_fused import cluster_stages_train, ctm_stage_group
        b = text_feat.shape[0] // world
        rows = slice(rank * b, (rank + 1) * b)
        tf, vf = text_feat[rows].contiguous(), video_feat[rows].contiguous()
        tm, vm = text_mask[rows].float().contiguous(), video_mask[rows].float().contiguous()
        n = {k: (v[rows].contiguous() if v is not None else None) for k, v in nz.items()}

        def exchange(smax):
            g = torch.stack([s_.max() for s_ in smax])
            comm.all_reduce(g, op="max")
            for s_, v in zip(smax, g):
                s_[:1] = v
        mods0 = ((self.text_ctm0, self.text_block0), (self.video_ctm0, self.video_block0))
        mods1 = ((self.text_ctm1, self.text_block1), (self.video_ctm1, self.video_block1))
        if torch.is_grad_enabled():
            t, v = cluster_stages_train(mods0, self._ctm_cache, ("text0", "video0"), tf, tm, n.get("t0"), vf, vm, n.get("v0"),
                                        exchange=exchange)
            return cluster_stages_train(mods1, self._ctm_cache, ("text1", "video1"), t, None, n.get("t1"), v, None, n.get("v1"))
        t, v = ctm_stage_group([("text0", tf, tm, self.text_ctm0, self.text_block0, n.get("t0")),
                                ("video0", vf, vm, self.video_ctm0, self.video_block0, n.get("v0"))], self._ctm_cache,
                               exchange=exchange)
        t, v = ctm_stage_group([("text1", t, None, self.text_ctm1, self.text_block1, n.get("t1")),
                                ("video1", v, None, self.video_ctm1, self.video_block1, n.get("v1"))], self._ctm_cache)
        return t, v

    @staticmethod
    def _gather_global(gt, gv, world):
        """The ranks' global tokens (equal shapes for text and video: [b,c,d]) in ONE all-gather -> ([B,c,d], [B,c,d])."""
        from . import comm
        if gt.shape != gv.shape:
            out = []
            for g in (gt, gv):
                full = torch.empty((world,) + tuple(g.shape), dtype=g.dtype, device=g.device)
                comm.all_gather_into_tensor(full.view(-1), g.contiguous().view(-1))
                out.append(full.flatten(0, 1))
            return tuple(out)
        pair = torch.stack((gt, gv)).contiguous()                                  # [2,b,c,d]
        full = torch.empty((world,) + tuple(pair.shape), dtype=pair.dtype, device=pair.device)
        comm.all_gather_into_tensor(full.view(-1), pair.view(-1))
        return full[:, 0].flatten(0, 1), full[:, 1].flatten(0, 1)

    def _merge_one(self, which, feat, mask, noise0=None, noise1=None):
        """Two CTM + TCBlock stages of one modality: [B,N,d] -> [B,1,d] at the MSR-VTT token counts."""
        ctm0, blk0 = getattr(self, which + "_ctm0"), getattr(self, which + "_block0")
        ctm1, blk1 = getattr(self, which + "_ctm1"), getattr(self, which + "_block1")
        if self._can_fuse_clustering(feat, (ctm0, blk0, ctm1, blk1)):
            from .cluster_fused import ctm_stage_fused
            t = ctm_stage_fused(feat, mask, ctm0, blk0, noise0, self._ctm_cache, which + "0")
            return ctm_stage_fused(t, None, ctm1, blk1, noise1, self._ctm_cache, which + "1")
        t = blk0(ctm0({"x": feat, "mask": mask.detach()}, noise0))
        return blk1(ctm1(t, noise1))["x"]

    def _can_fuse_clustering(self, feat, mods):
        """Fused forward kernels when no gradient is wanted; autograd-traced torch ops otherwise."""
        if not (feat.is_cuda and self.fuse_clustering and feat.shape[1] <= 64 and feat.shape[2] % 64 == 0):
            return False
        if not torch.is_grad_enabled():
            return True
        return not (feat.requires_grad or any(p.requires_grad for m in mods for p in m.parameters()))

    @staticmethod
    def _token_counts(Nt, Nv):
        """((text tokens after stage 0, after stage 1), (video ...)): cluster.py:712 with the ratios of
        modeling.py:188-196 (the same float products as the CTM modules evaluate)."""
        t0 = max(math.ceil(Nt * (1 / 6)), 1)
        v0 = max(math.ceil(Nv * (1 / 4)), 1)
        return (t0, max(math.ceil(t0 * (1 / 4)), 1)), (v0, max(math.ceil(v0 * (1 / 3)), 1))

    @classmethod
    def _noise_sizes(cls, Nt, Nv):
        (t0, _), (v0, _) = cls._token_counts(Nt, Nv)
        return {"t0": Nt, "t1": t0, "v0": Nv, "v1": v0}

    @staticmethod
    def _slice_noise(flat, B, sizes):
        out, off = {}, 0
        for k, n in sizes.items():
            out[k] = flat[off:off + B * n].view(B, n)
            off += B * n
        return out

    @classmethod
    def _draw_noise(cls, B, Nt, Nv, device):
        """The four DPC-KNN tie-break draws of one step (cluster.py:483) from ONE torch.rand launch."""
        sizes = cls._noise_sizes(Nt, Nv)
        return cls._slice_noise(torch.rand(B * sum(sizes.values()), device=device, dtype=torch.float32), B, sizes)

    def _rng_state_on(self, device):
        """Device-resident {seed, counter} of the step prologue's noise stream (seeded from torch's seed)."""
        if self._rng_state is None or self._rng_state.device != device:
            self._rng_state = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=device)
        return self._rng_state

    # Side streams (neighborretr_amd.streams): eager steps use streams cached per model; inside a HIP-graph capture every name
    # resolves to a stream created for THAT capture alone -- a stream object that has been part of an earlier capture is never
    # handed to a later one (the re-captures of main_retrieval.GraphedStep, one per bank generation, walk this path).
    def _slot(self):
        return self._pipeline.slot if self._pipeline is not None else 0

    def _head_slot(self, slot, like):
        slots = self.__dict__.setdefault("_head_slots", {})
        key = (int(slot), like.device)
        if key not in slots:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("pipelined steps: call model.prepare_pipeline(n_steps, batch) before the capture (the per-step "
                                   "copies of the ring head must not live in a graph's memory pool)")
            slots[key] = torch.zeros_like(like)
        return slots[key]

    def prepare_pipeline(self, n_steps, batch_rows):
        """Before capturing `n_steps` overlapped steps (StepPipeline): the persistent per-step words they need."""
        ring = self._ring_ready(batch_rows)
        if ring is not None:
            for k in range(int(n_steps)):
                self._head_slot(k, ring[0])
            for k in range(int(n_steps)):          # (allocated inside the capture, each would cost a fill node on the origin stream)
                ops.split_tail_counter(ring[0].device, k)
        from . import streams
        streams.reserve(8 * int(n_steps) + 8)      # a pipelined step forks up to seven streams: none is created while the capture is open

    def _pipeline_for_head(self, device):
        """The installed StepPipeline with its tail stream resolved (None: steps are not being captured overlapped)."""
        pipe = self._pipeline
        if pipe is not None and pipe.tail_stream is None:
            from . import streams
            pipe.tail_stream = streams.side(self, self._sname("tail"), device)
        return pipe

    def _sname(self, name):
        """Stream names of the running step: the steps of a pipelined capture (StepPipeline) each fork their own set."""
        return name if self._pipeline is None else f"{name}#{self._pipeline.slot}"

    def _side_streams(self, device):
        # default priority: high-priority side streams made the captured graph 1.8x SLOWER on ROCm 7.2
        from . import streams
        return streams.side(self, self._sname("text"), device), streams.side(self, self._sname("video"), device)

    def _cluster_stream(self, device):
        """Stream of the training step's token clustering (forward and, through autograd, backward), or None."""
        if not (self.use_side_streams and self.cluster_side_stream and device.type == "cuda"):
            return None
        from . import streams
        return streams.side(self, self._sname("cluster"), device)

    def _local_stream(self, device):
        if not (self.use_side_streams and device.type == "cuda"):
            return None
        from . import streams
        return streams.side(self, self._sname("local"), device)

    def _bank_streams(self, device):
        """Side streams of the two memory-bank chains (head.head_forward), or None when streams are off."""
        if not (self.use_side_streams and self.bank_side_streams and device.type == "cuda"):
            return None
        from . import streams
        return tuple(streams.side(self, self._sname(n), device) for n in ("bank0", "bank1", "push"))      # two bank chains + the bank push

    def _take_join(self):
        j, self._join_global = self._join_global, None
        return j

    # ------------------------------------------------------------------ similarity API
    def local_level(self, text_feat, video_feat, text_mask, video_mask):
        """(S, S.T) with S [A,Bv] -- modeling.py:483-514, fused on the GPU; differentiable."""
        from .functional import local_level_sim
        S = local_level_sim(self, text_feat, video_feat, text_mask, video_mask)
        return S, S.T

    def global_level(self, text_feat, video_feat):
        """modeling.py:516-539."""
        from .functional import global_level_sim
        G = global_level_sim(self, text_feat, video_feat)
        return G, G.T

    def get_similarity_logits(self, text_feat, video_feat, text_mask, video_mask, shaped=False):
        if shaped is False:
            text_mask = text_mask.view(-1, text_mask.shape[-1])
            video_mask = video_mask.view(-1, video_mask.shape[-1])
        S, _ = self.local_level(text_feat, video_feat, text_mask, video_mask)
        return S, S.T

    # ------------------------------------------------------------------ feature producers
    def get_text_feat(self, text_ids, text_mask, shaped=False):
        if self.feature_mode:
            return torch.as_tensor(text_ids).float().view(-1, text_ids.shape[-2], text_ids.shape[-1])
        if shaped is False:
            text_ids = text_ids.view(-1, text_ids.shape[-1])
            text_mask = text_mask.view(-1, text_mask.shape[-1])
        bs = text_ids.size(0)
        with self._encoder_autocast(text_ids.device):
            _, feat = self.clip.encode_text(text_ids, return_hidden=True, mask=text_mask)
        return feat.float().view(bs, -1, feat.size(-1))

    def _encoder_autocast(self, device):
        on = device.type == "cuda" and self.encoder_dtype is not None
        return torch.autocast("cuda", dtype=self.encoder_dtype, enabled=on)

    def get_video_feat(self, video, video_mask, shaped=False):
        """modeling.py:553-567: frames through the image tower (class token of every frame), then the temporal
        transformer over the frames of a video (aggregate_video_features, :601-623)."""
        if self.feature_mode:
            return torch.as_tensor(video).float().view(-1, video.shape[-2], video.shape[-1])
        if not hasattr(self, "transformerClip"):
            raise NotImplementedError("pixel input needs the temporal transformer: construct NeighborRetr(config, "
                                      "with_encoders=True) (or subclass with your own get_video_feat)")
        if shaped is False:
            video_mask = video_mask.view(-1, video_mask.shape[-1])
            video = torch.as_tensor(video)
            video = video.reshape((-1,) + tuple(video.shape[-3:]))          # [b, n_v, 3, H, W] or the 7-D loader layout
        bs, n_v = video_mask.shape
        from .encoders import aggregate_video_features
        with self._encoder_autocast(video.device):
            frame = self.clip.encode_image(video.to(self.clip.visual.conv1.weight.dtype), return_hidden=True)[0]
            frame = frame.float().view(bs, -1, frame.size(-1))
            feat = aggregate_video_features(frame, video_mask, self.frame_position_embeddings, self.transformerClip)
        return feat.float()

    def get_text_video_feat(self, text_ids, text_mask, video, video_mask, shaped=False):
        if not self.feature_mode and shaped is False:
            text_ids = text_ids.view(-1, text_ids.shape[-1])
            text_mask = text_mask.view(-1, text_mask.shape[-1])
            video_mask = video_mask.view(-1, video_mask.shape[-1])
            video = torch.as_tensor(video)
            video = video.reshape((-1,) + tuple(video.shape[-3:]))
            shaped = True
        return self.get_text_feat(text_ids, text_mask, shaped), self.get_video_feat(video, video_mask, shaped)


def default_config(**over):
    return SimpleNamespace(**{**DEFAULTS, **over})
