#!/usr/bin/env python
"""DDP entry point for the similarity / loss path (the reference's main.py:189-437, launched in its
README as main_retrieval.py).  Same flag names, same per-epoch sequence
  load memory bank -> train epoch -> eval -> save -> clear bank,
same training-step contract (trainer.py:66-203): model(...) -> 5 losses -> backward ->
clip_grad_norm_(1.0) -> optimizer step -> clamp logit_scale <= ln 100 -> reduce the 5 scalars to
rank 0.  One process per GPU (torchrun / torch.distributed.run; RCCL is torch's "nccl" backend).

The encoders and the video/text datasets are out of this build's scope, so the
runnable mode is `--synthetic`: seeded CLIP-shaped token features stand in for the encoder outputs
and the model runs in feature mode.  With real encoders, construct
`NeighborRetr(args, clip=<module with encode_text/encode_image/logit_scale>)` instead.

  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 main_retrieval.py \\
      --do_train 1 --synthetic --batch_size 128 --max_words 24 --max_frames 12 --mb_batch 4 --epochs 1
"""
import argparse
import os
import time

import numpy as np
import torch
import torch.distributed as dist


def get_args():
    p = argparse.ArgumentParser("NeighborRetr on MI355X")
    # loss parameters (args_parser.py:25-41)
    p.add_argument("--centrality_scale", default=0.3, type=float)
    p.add_argument("--kl_weight", default=1.0, type=float)
    p.add_argument("--uniform_weight", default=1.0, type=float)
    p.add_argument("--ot_temperature", default=0.1, type=float, help="parsed and unused, as in the reference")
    p.add_argument("--beta", default=0.7, type=float)
    p.add_argument("--num_neighbors", default=20, type=int)
    p.add_argument("--temperature", default=3.0, type=float)
    p.add_argument("--neighbor_weight", default=1.0, type=float)
    # data loading / modes / dataset (accepted for command-line compatibility)
    p.add_argument("--workers", default=8, type=int)
    p.add_argument("--pin_memory", action="store_true")
    p.add_argument("--prefetch_factor", default=4, type=int)
    p.add_argument("--persistent_workers", action="store_true")
    p.add_argument("--video_cache_size", default=64, type=int)
    p.add_argument("--use_prefetch", action="store_true")
    p.add_argument("--timeout", default=0, type=int)
    p.add_argument("--save_model", action="store_true")
    p.add_argument("--do_train", type=int, default=0)
    p.add_argument("--do_eval", type=int, default=0)
    p.add_argument("--detect_grad", action="store_true")
    p.add_argument("--datatype", default="msrvtt", type=str)
    p.add_argument("--anno_path", type=str, default="data/MSR-VTT/anno")
    p.add_argument("--video_path", type=str, default="data/MSR-VTT/videos")
    p.add_argument("--output_dir", default="output", type=str)
    p.add_argument("--seed", type=int, default=42)
    # optimisation
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--coef_lr", type=float, default=1e-3)
    p.add_argument("--warmup_proportion", default=0.1, type=float)
    p.add_argument("--weight_decay", type=float, default=0.2)
    p.add_argument("--epochs", type=int, default=5)
    # batch
    p.add_argument("--batch_size", type=int, default=128, help="GLOBAL batch (data_dataloaders.py:38)")
    p.add_argument("--batch_size_val", type=int, default=128)
    p.add_argument("--memory_size", type=int, default=512, help="parsed and unused, as in the reference")
    p.add_argument("--mb_batch", type=int, default=10)
    p.add_argument("--max_words", type=int, default=24)
    p.add_argument("--max_frames", type=int, default=12)
    p.add_argument("--video_framerate", type=int, default=1)
    # distributed / model
    p.add_argument("--device", default="cpu", type=str)
    p.add_argument("--world_size", default=1, type=int)
    p.add_argument("--local_rank", "--local-rank", default=0, type=int)
    p.add_argument("--distributed", default=0, type=int)
    p.add_argument("--n_display", type=int, default=50)
    p.add_argument("--base_encoder", default="ViT-B/32", type=str)
    p.add_argument("--num_hidden_layers", type=int, default=4)
    p.add_argument("--init_model", default=None, type=str)
    # this build
    p.add_argument("--synthetic", action="store_true", help="seeded token features instead of encoders + datasets")
    p.add_argument("--synthetic_train", type=int, default=2048, help="synthetic training pairs")
    p.add_argument("--synthetic_test", type=int, default=1000, help="synthetic test pairs (MSR-VTT 1k-A size)")
    p.add_argument("--precision", default="bf16", choices=["bf16", "bf16x3", "bf16_all"])
    p.add_argument("--dist_backend", default="nccl", choices=["nccl", "gloo"],
                   help="collective backend: nccl = RCCL over xGMI (one GPU per rank); gloo only to rehearse several "
                        "ranks on ONE GPU (tests)")
    p.add_argument("--centrality_multi_token", default="raise", choices=["raise", "mean"],
                   help="several global tokens per sample (ActivityNet token counts): 'raise' like the reference "
                        "(until_module.py:321), 'mean' = centrality weight averaged over the sample's global tokens")
    p.add_argument("--encoders", type=int, default=0,
                   help="1: BASELINE configs[4] -- ViT-B/32 image tower, text tower and temporal transformer (stock "
                        "PyTorch-ROCm modules, random init unless --init_model) in front of the HIP head; --synthetic then "
                        "feeds random pixels [b, frames, 3, 224, 224] and token ids instead of token features")
    p.add_argument("--hip_graph", type=int, default=0,
                   help="1: the training step replayed from captured HIP graphs instead of ~90 eager launches.  One rank: forward + "
                        "backward as ONE graph.  Several ranks: the whole data-parallel step -- exchange, loss, backward, gradient "
                        "average -- as one graph with the RCCL collectives inside, or (when that is refused / fails its validation, "
                        "and on gloo) the rank-local segments between the collectives as graphs; the ranks decide together")
    args = p.parse_args()
    if args.batch_size % max(1, int(os.environ.get("WORLD_SIZE", "1"))):
        raise ValueError("--batch_size must divide over the ranks (args_parser.py:149-165)")
    return args


def setup_distributed(args):
    """One process per GPU; rendezvous from the torchrun environment (setup.py:44-69)."""
    args.world_size = int(os.environ.get("WORLD_SIZE", "1"))
    args.rank = int(os.environ.get("RANK", "0"))
    args.local_rank = int(os.environ.get("LOCAL_RANK", str(args.local_rank)))
    if not torch.cuda.is_available():
        raise RuntimeError("the HIP path needs an MI355X; there is no CPU fallback")
    n_dev = torch.cuda.device_count()
    if args.dist_backend == "nccl" and args.local_rank >= n_dev:
        raise RuntimeError(f"LOCAL_RANK {args.local_rank} but {n_dev} GPU(s) visible: RCCL needs one GPU per rank")
    args.device_index = args.local_rank % n_dev                     # gloo rehearsal: ranks share the card
    torch.cuda.set_device(args.device_index)
    args.device = torch.device("cuda", args.device_index)
    if args.world_size > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=args.device)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")
    # AllGather / packed_allgather slice gradients by GLOBAL rank
    args.local_rank = args.rank
    return args


def log(args, msg):
    if args.rank == 0:
        print(time.strftime("%H:%M:%S"), msg, flush=True)


class SyntheticFeatures:
    """Per-rank shard of seeded (text, video) token features: the encoders' stand-in."""

    def __init__(self, args, n, stream, seed):
        from neighborretr_amd import synth
        t, v, tm, vm = synth.make_samples(seed, stream, n, args.max_words, args.max_frames)
        self.t, self.v, self.tm, self.vm = (torch.from_numpy(a) for a in (t, v, tm, vm))
        self.n = n
        self.b = args.batch_size // args.world_size
        self.W, self.rank, self.B = args.world_size, args.rank, args.batch_size

    def __len__(self):
        return self.n // self.B

    def batch(self, i, device):
        lo = i * self.B + self.rank * self.b
        sl = slice(lo, lo + self.b)
        idx = torch.arange(lo, lo + self.b)
        return tuple(x.to(device, non_blocking=True) for x in (self.t[sl], self.tm[sl], self.v[sl], self.vm[sl], idx))


class SyntheticClips:
    """Per-rank shard of seeded raw inputs for --encoders 1: token ids [b, Nt] (EOT at the end of the mask) and pixels
    [b, Nv, 3, 224, 224], generated on the device batch by batch (a whole epoch of frames would not fit the host)."""

    def __init__(self, args, n, stream, seed, resolution=224):
        from neighborretr_amd import synth
        from neighborretr_amd.encoders import synthetic_text_ids
        _, _, tm, vm = synth.make_samples(seed, stream, n, args.max_words, args.max_frames, d=8)
        self.tm, self.vm = torch.from_numpy(tm), torch.from_numpy(vm)
        self.ids = synthetic_text_ids(self.tm, seed=seed)
        self.n, self.res, self.seed = n, resolution, seed
        self.b = args.batch_size // args.world_size
        self.W, self.rank, self.B = args.world_size, args.rank, args.batch_size

    def __len__(self):
        return self.n // self.B

    def rows(self, index, device):
        g = torch.Generator(device=device).manual_seed(self.seed * 1000003 + int(index[0]))
        video = torch.randn((len(index), self.vm.shape[1], 3, self.res, self.res), generator=g, device=device)
        return (self.ids[index].to(device), self.tm[index].to(device), video, self.vm[index].to(device), index.to(device))

    def batch(self, i, device):
        lo = i * self.B + self.rank * self.b
        return self.rows(torch.arange(lo, lo + self.b), device)


def encode(model, batch):
    """(text_feat, text_mask, video_feat, video_mask, idx) of a raw batch: the encoders' outputs, or the inputs themselves
    in feature mode."""
    text, tm, video, vm, idx = batch
    if model.feature_mode:
        return batch
    with torch.no_grad():
        tf, vf = model.get_text_video_feat(text, tm, video, vm)
    return tf, tm, vf, vm, idx


def load_memory_bank(args, model, data):
    """memory_bank.py:80-229: run mb_batch batches through the (feature-mode) encoders under no_grad,
    gather them over the ranks, hand them to the model as plain attributes."""
    from neighborretr_amd.dist import packed_allgather
    n = min(args.mb_batch, len(data))
    was_training = model.training
    model.eval()                                            # memory_bank.py:102
    feats = [encode(model, data.batch(i, args.device)) for i in range(n)]
    model.train(was_training)
    with torch.no_grad():
        tf, tm, vf, vm, idx = (torch.cat([f[k] for f in feats], 0) for k in range(5))
        tf, vf, idx, tm, vm = packed_allgather(tf, vf, idx, tm, vm, args)
    model.mb_ind, model.mb_feat_t, model.mb_feat_v = idx, tf.contiguous(), vf.contiguous()
    model.mb_mask_t, model.mb_mask_v, model.mb_batch = tm.contiguous(), vm.contiguous(), tf.shape[0]   # memory_bank.py:211
    log(args, f"memory bank: {tf.shape[0]} samples ({n} batches x {args.batch_size})")
    return tf.shape[0]


def clear_memory_bank(model):
    model._init_memory_bank()


class GraphedStep:
    """Forward + backward of one training step as a captured HIP graph with static input buffers: run() leaves the losses in
    .losses and the gradients in the parameters' .grad (the optimizer step stays eager).

    world_size > 1: the step that is replayed is the WHOLE data-parallel step -- exchange step, loss, backward with the
    reductions of its differentiable collectives, and the gradient average over the ranks (one all-reduce of a flat buffer that
    the parameters' .grad are views of; what DistributedDataParallel's bucketed all-reduce computes, optimizer.py:79-84) --
    in the best form every rank can take (neighborretr_amd.comm.CollectiveCapture): ONE graph with the RCCL collectives
    inside; else the rank-local segments between the collectives as graphs (comm.SegmentedStep); else eager launches.  Each
    form is validated against the eager step on every rank before it is used."""

    def __init__(self, model, example, params, args=None):
        self.static = [t.clone() for t in example]
        self.params = params
        self.model = model
        self.world = int(getattr(args, "world_size", 1)) if args is not None else 1
        self.rank = int(getattr(args, "rank", 0)) if args is not None else 0
        self.backend = getattr(args, "dist_backend", "nccl") if args is not None else "nccl"
        self.form = "eager"
        self.cc = None
        self.capture()

    def capture(self):
        with self.model.graph_capture_mode():
            if self.world > 1:
                self._capture_parallel()
            else:
                self._capture()

    def _warm_up(self, step):
        """Warm-up on a side stream, as graph capture requires -- with the bank FROZEN: a warm-up step that pushed its batch
        would leave the static batch in the bank three times (and three more times after every re-capture), which is not the
        reference's FIFO (modeling.py:222-249)."""
        model = self.model
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        model.bank_frozen = True
        try:
            with torch.cuda.stream(side):
                for _ in range(3):
                    step()
        finally:
            model.bank_frozen = False
        torch.cuda.current_stream().wait_stream(side)

    def _capture(self):
        model, params = self.model, self.params
        B = self.static[0].shape[0]

        def fwd_bwd():
            return torch.autograd.grad(model(*self.static, 0)[0], params, allow_unused=True)
        self._warm_up(fwd_bwd)
        # the device-resident ring head must exist before the capture (creating it is a host-to-device copy)
        model._ring_ready(B)
        torch.cuda.synchronize()
        for p in params:
            p.grad = None
        # derived weights (bf16 hi/lo splits) are cached per parameter version: drop the caches so that the splits are
        # captured too and every replay re-derives them from the fp32 parameters the optimizer has just updated
        model._scorer_cache.clear()
        model._ctm_cache.clear()
        # torch.autograd.grad, not .backward(): the gradients come back as the graph's own static tensors and NO AccumulateGrad
        # node takes part.  Those nodes outlive a step whenever anything still holds its losses (a training loop's `losses`
        # variable, DDP), on the stream they were created on; a capture that has to synchronise with such a stream -- the
        # default stream in particular -- dies inside the runtime (PyTorch warns "AccumulateGrad node's stream does not match
        # ... break CUDA graph capture"; measured: tools/rank_local_times.py, round 4).
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            losses = model(*self.static, 0)
            grads = torch.autograd.grad(losses[0], params, allow_unused=True)
        self.losses = tuple(l.detach() for l in losses)          # the loss VALUES only (no autograd graph kept alive)
        del losses
        self.grads = list(grads)                     # static gradient buffers of the graph (None: the step does not reach it)
        self.replay, self.form = self.graph.replay, "whole"
        self._remember_bank()

    def _remember_bank(self):
        # What the graph has baked in: the addresses of the five bank tensors and of the device ring head.  Hold
        # strong references (a bank replaced from outside must not hand its blocks to somebody else while this graph
        # can still replay) and remember the bank's storage generation; run() re-captures when it has moved on.
        model = self.model
        self.bank_refs = (dict(model._mb), model._mb_head_dev)
        self.generation = model._mb_gen

    def _capture_parallel(self):
        from neighborretr_amd import comm
        model, params, W = self.model, self.params, self.world
        B = self.static[0].shape[0] * W
        if self.cc is None:
            self.cc = comm.CollectiveCapture(W, self.rank, log=lambda msg: print(f"[GraphedStep] {msg}", flush=True))

        seen = {}

        def fwd_bwd():
            seen["g"] = torch.autograd.grad(model(*self.static, 0)[0], params, allow_unused=True)
        self._warm_up(fwd_bwd)
        model._ring_ready(B)
        torch.cuda.synchronize()
        # the parameters that receive a gradient (the same set in every step: DDP's find_unused_parameters bookkeeping,
        # optimizer.py:79-84, done once) get views of ONE flat buffer as their .grad
        used = [p for p, g in zip(params, seen.pop("g")) if g is not None]
        flat = torch.zeros(sum(p.numel() for p in used), dtype=torch.float32, device=used[0].device)
        views, off = [], 0
        for p in used:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        out = {}
        rng = model._rng_state_on(flat.device)

        def step():
            losses = model(*self.static, 0)
            grads = torch.autograd.grad(losses[0], used)          # (no AccumulateGrad nodes in a captured step: see _capture)
            torch._foreach_copy_(views, list(grads))
            comm.all_reduce(flat)                     # the gradient average over the ranks: one collective, one flat buffer
            flat.mul_(1.0 / W)
            out["losses"] = torch.stack([l.detach() for l in losses])

        def eager_pass():
            rng[1] = 4242                             # the same DPC-KNN tie-break draws for the eager and the replayed pass
            step()

        def result():
            return [out["losses"], flat]

        def same(a, b_):
            return bool(torch.allclose(a[0], b_[0], rtol=1e-4, atol=1e-6)
                        and (a[1] - b_[1]).norm() <= 1e-3 * b_[1].norm() + 1e-12)

        def whole():
            model._scorer_cache.clear()
            model._ctm_cache.clear()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                step()

            def replay_pass():
                rng[1] = 4242
                g.replay()
            return replay_pass, g

        def segmented():
            model._scorer_cache.clear()
            model._ctm_cache.clear()
            seg = comm.SegmentedStep(step, capture_error_mode="relaxed").capture()

            def replay_pass():
                rng[1] = 4242
                seg.replay()
            return replay_pass, seg
        freeze = lambda on: setattr(model, "bank_frozen", on)      # noqa: E731
        # the validation passes pin the DPC-KNN noise counter to one value; the run continues from where it stood before them
        # (otherwise every re-capture -- one per epoch -- restarts the same noise sequence and a --hip_graph 1 run diverges
        # from the eager run with the same seed)
        rng_before = rng.clone()
        form = self.cc.attempt("whole-step", eager_pass, whole, result, same, freeze) if self.backend == "nccl" else None
        self.form = "whole"
        if form is None:
            form = self.cc.attempt("segmented", eager_pass, segmented, result, same, freeze)
            self.form = "segmented"
        rng.copy_(rng_before)
        if form is None:
            self.form, self.replay, self.keep = "eager", step, None
        else:
            self.keep = form[1]
            self.replay = form[1].replay
        self.grads = None
        self._used, self._views, self._out = used, views, out
        self._remember_bank()

    def _eager(self, batch):
        """A batch whose shapes differ from the captured ones (a loader's last, shorter batch): one eager step with the same
        outcome -- losses returned, gradients left in .grad -- instead of a graph per shape."""
        if self.world > 1:
            raise ValueError("GraphedStep on several ranks needs batches of one shape (the exchange step gathers equal shards: drop "
                             f"the loader's last batch); got {[tuple(t.shape) for t in batch]} after {[tuple(t.shape) for t in self.static]}")
        losses = self.model(*batch, 0)
        grads = torch.autograd.grad(losses[0], self.params, allow_unused=True)
        for p, g in zip(self.params, grads):
            p.grad = g
        return tuple(l.detach() for l in losses)

    def run(self, batch):
        if any(tuple(src.shape) != tuple(dst.shape) for dst, src in zip(self.static, batch)):
            return self._eager(batch)
        if self.model._mb_gen != self.generation:    # the bank's tensors / ring head were replaced: the graph is stale
            self.capture()
        for dst, src in zip(self.static, batch):
            dst.copy_(src)
        self.replay()
        if self.world > 1:
            for p, g in zip(self._used, self._views):    # optimizer.zero_grad(set_to_none=True) drops them: put them back
                p.grad = g
            return tuple(self._out["losses"].unbind(0))
        for p, g in zip(self.params, self.grads):    # optimizer.zero_grad(set_to_none=True) drops them: put them back
            p.grad = g
        return self.losses


def train_epoch(args, model, ddp, data, optimizer, epoch, global_step):
    from neighborretr_amd.dist import reduce_losses
    model.train()
    t0 = time.time()
    graphed = getattr(args, "_graphed_step", None)
    for i in range(len(data)):
        global_step += 1
        text, text_mask, video, video_mask, idx = data.batch(i, args.device)
        if args.hip_graph:
            if graphed is None:
                graphed = args._graphed_step = GraphedStep(model, (text, text_mask, video, video_mask, idx),
                                                           [p for p in model.parameters() if p.requires_grad], args)
                log(args, f"training step replayed as: {graphed.form}")
            losses = graphed.run((text, text_mask, video, video_mask, idx))
            loss = None
        else:
            losses = ddp(text, text_mask, video, video_mask, idx, global_step)
            loss = losses[0]
            loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        optimizer.step()
        optimizer.zero_grad(set_to_none=True)
        torch.clamp_(model.clip.logit_scale.data, max=float(np.log(100)))        # trainer.py:114-119
        if global_step % args.n_display == 0 or i == len(data) - 1:
            red = reduce_losses(losses, args).tolist()                            # one reduce instead of five
            log(args, f"epoch {epoch} step {i + 1}/{len(data)} loss {red[0]:.4f} centrality {red[1]:.4f} "
                      f"uniform {red[2]:.4f} neighbor {red[3]:.4f} kl {red[4]:.4f} "
                      f"({(time.time() - t0) / (i + 1) * 1e3:.1f} ms/step)")
    return global_step


def eval_epoch(args, model, test):
    """evaluator.py:66-291 for the single-sentence case with the work SHARDED over the ranks (neighborretr_amd.evaluator):
    every rank "extracts" the features of its samples (rank, rank + W, ...: a DistributedSampler's split; feature mode:
    they are the inputs), one packed all-gather + index scatter restores dataset order (evaluator.py:173-189), rank r
    computes rows [r N/W, (r+1) N/W) of the N x N similarity and the rank counts of its slab on the GPU, three small
    collectives complete them."""
    from neighborretr_amd.evaluator import gather_eval_features, rank_sample_indices, sharded_metrics
    model.eval()
    dev = args.device
    mine = rank_sample_indices(test.n, args.world_size, args.rank)     # equal counts on every rank (padded like DistributedSampler)
    if model.feature_mode:
        t, tm, v, vm = (x[mine].to(dev) for x in (test.t, test.tm, test.v, test.vm))
    else:                                                   # evaluator.py:162-171: features cached batch by batch
        parts = [encode(model, test.rows(mine[lo:lo + 32], dev)) for lo in range(0, len(mine), 32)]
        t, tm, v, vm = (torch.cat([p[k] for p in parts], 0) for k in range(4))
    if args.world_size > 1:
        t, v, tm, vm = gather_eval_features(t, v, mine.to(dev), tm, vm, args)
    t2v, v2t = sharded_metrics(model, t, v, tm.float(), vm.float(), args)
    log(args, f"text->video R@1 {t2v['R1']:.1f} R@5 {t2v['R5']:.1f} R@10 {t2v['R10']:.1f} MedR {t2v['MR']:.1f} | "
              f"video->text R@1 {v2t['R1']:.1f} R@5 {v2t['R5']:.1f} R@10 {v2t['R10']:.1f} MedR {v2t['MR']:.1f}")
    return t2v, v2t


def main():
    args = get_args()
    args = setup_distributed(args)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    if not args.synthetic:
        raise SystemExit("datasets / CLIP towers are outside this build (SURVEY.md 2.1): run with --synthetic, "
                         "or import neighborretr_amd.modeling.NeighborRetr into the reference's main.py")
    from neighborretr_amd.modeling import NeighborRetr
    model = NeighborRetr(args, precision=args.precision, with_encoders=bool(args.encoders))
    if args.init_model:
        sd = torch.load(args.init_model, map_location="cpu")
        missing, unexpected = model.load_state_dict(sd, strict=False)          # main.py:60-67
        log(args, f"init_model: {len(missing)} missing / {len(unexpected)} unexpected keys")
    model = model.to(args.device)
    ddp = model
    # (--hip_graph 1 replays the whole data-parallel step, gradient average included, from graphs: GraphedStep; a DDP wrapper
    # would keep AccumulateGrad nodes of the default stream alive, which a capture must not be synchronised with)
    if args.world_size > 1 and not args.hip_graph:
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[args.device_index],
                                                        find_unused_parameters=True)   # optimizer.py:79-84
    Data = SyntheticClips if args.encoders else SyntheticFeatures
    train = Data(args, args.synthetic_train, "train", args.seed)
    test = Data(args, args.synthetic_test, "test", args.seed + 1)
    optimizer = torch.optim.AdamW(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)
    os.makedirs(args.output_dir, exist_ok=True)
    global_step = 0
    if args.do_train:
        for epoch in range(args.epochs):
            load_memory_bank(args, model, train)
            global_step = train_epoch(args, model, ddp, train, optimizer, epoch + 1, global_step)
            eval_epoch(args, model, test)
            if args.rank == 0 and args.save_model:
                torch.save(model.state_dict(), os.path.join(args.output_dir, f"pytorch_model.bin.{epoch}"))
            clear_memory_bank(model)
    elif args.do_eval:
        eval_epoch(args, model, test)
    if args.world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
