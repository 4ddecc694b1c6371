"""The callers on either side of the loss head, with the reference's signatures:

    MemoryBankManager        NeighborRetr/utils/memory_bank.py:22-260
    eval_epoch               NeighborRetr/training/evaluator.py:66-291   (single- and multi-sentence test sets)
    train_epoch              NeighborRetr/training/trainer.py:18-221
    reduce_loss              NeighborRetr/utils/setup.py:72-94

so that the reference's main.py runs on this package by changing its imports.  They take what the reference's take: a
DataLoader-like iterable of `(text_ids, text_mask, video, video_mask, inds, idx)` batches, `args` with `logger`, `world_size`,
`rank`/`local_rank`, `n_display`, `epochs`, ... -- and differ from it in how the work is laid out, not in what comes back:

  * the exchange steps are the packed collective of neighborretr_amd.dist (one all-gather instead of five + a barrier);
  * evaluation never builds the N x N matrix on every rank, nor copies 64 x 64 tiles to the host: each rank scores its
    row slab with the fused split-bf16 kernel and counts ranks on the GPU (neighborretr_amd.evaluator);
  * the five logged losses are reduced with one collective instead of five;
  * the memory bank handed to the model is adopted by its ring + bf16 shadow (modeling._bank_set), not re-read per step.

Datasets, tokenizer, BertAdam and checkpoints stay the reference's own (out of scope): any torch optimizer with
`step()/zero_grad()` works here; `get_lr()` is used for the log line when the optimizer has it.
"""
import logging
import time
from datetime import timedelta

import numpy as np
import torch
import torch.distributed as dist

from .metrics import RetrievalMetrics


def is_main_process():
    """utils/comm.py: rank 0, or no process group at all."""
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def _info(logger, msg):
    if logger is not None and is_main_process():
        logger.info(msg)


def reduce_loss(loss, args):
    """setup.py:72-94: mean over the ranks, valid on rank 0 (the other ranks keep their partial sums, as there)."""
    if int(getattr(args, "world_size", 1)) < 2:
        return loss
    with torch.no_grad():
        dist.reduce(loss, dst=0)
        if dist.get_rank() == 0:
            loss /= args.world_size
    return loss


def _gather(tensors, args):
    """[text_feat, video_feat, idx, text_mask, video_mask] of every rank, rows concatenated in rank order."""
    if int(getattr(args, "world_size", 1)) < 2 or not dist.is_initialized():
        return tensors
    from .dist import packed_allgather
    with torch.no_grad():
        return packed_allgather(*tensors, args)


class MemoryBankManager:
    """memory_bank.py:22-260.  `create_memory_bank_dataloader` belongs to the reference's dataloaders (out of scope here):
    hand the training DataLoader to `load_memory_bank`, or set `manager.memory_bank_dataloader` once."""

    def __init__(self, args):
        self.args = args
        self.logger = getattr(args, "logger", None)
        self.mb_batch = getattr(args, "mb_batch", 10)
        self.batch_size = args.batch_size
        self.memory_bank_dataloader = None

    def create_memory_bank_dataloader(self):
        if self.memory_bank_dataloader is None:
            factory = getattr(self.args, "memory_bank_dataloader_factory", None)
            if factory is None:
                raise NotImplementedError("no dataset code in this package: set manager.memory_bank_dataloader (the training "
                                          "DataLoader) or args.memory_bank_dataloader_factory")
            self.memory_bank_dataloader = factory(self.args)
        return self.memory_bank_dataloader

    def load_memory_bank(self, model, memory_bank_dataloader, device, epoch):
        """Features of the first `mb_batch` batches under no_grad / eval mode, gathered over the ranks, handed to the model as
        its five bank attributes (memory_bank.py:80-229).  -> number of samples in the bank."""
        target = _unwrap(model).to(device)
        was_training = target.training
        target.eval()                                                           # memory_bank.py:102
        loader = memory_bank_dataloader if memory_bank_dataloader is not None else self.create_memory_bank_dataloader()
        n_batches = min(self.mb_batch, len(loader))
        _info(self.logger, f"Memory bank loading: target {n_batches} batches out of {len(loader)}")
        parts = {k: [] for k in ("ind", "tf", "tm", "vf", "vm")}
        with torch.no_grad():
            for batch_idx, batch in enumerate(loader):
                if batch_idx >= n_batches:
                    break
                text_ids, text_mask, video, video_mask, indices, _ = (t.to(device=device, non_blocking=True) for t in batch)
                tf, vf = target.get_text_video_feat(text_ids, text_mask, video, video_mask)
                parts["ind"].append(indices.reshape(-1))
                parts["tf"].append(tf.float())
                parts["vf"].append(vf.float())
                parts["tm"].append(text_mask.view(-1, text_mask.shape[-1]))
                parts["vm"].append(video_mask.view(-1, video_mask.shape[-1]))
        target.train(was_training)
        if not parts["ind"]:
            if self.logger is not None:
                self.logger.warning("No batches were processed for the memory bank")
            return 0
        ind, tf, tm, vf, vm = (torch.cat(parts[k], 0) for k in ("ind", "tf", "tm", "vf", "vm"))
        if getattr(self.args, "distributed", int(getattr(self.args, "world_size", 1)) > 1):
            tf, vf, ind, tm, vm = _gather([tf, vf, ind, tm, vm], self.args)
        target.mb_ind, target.mb_feat_t, target.mb_mask_t = ind, tf.contiguous(), tm.contiguous()
        target.mb_feat_v, target.mb_mask_v = vf.contiguous(), vm.contiguous()
        target.mb_batch = tf.size(0)
        gib = sum(t.numel() * t.element_size() for t in (ind, tf, tm, vf, vm)) / 2 ** 30
        _info(self.logger, f"Memory bank size: {tf.size(0)} samples, {gib:.3f} GB")
        _info(self.logger, f"Feature dimensions - Text: {tuple(tf.shape)}, Video: {tuple(vf.shape)}")
        return tf.size(0)

    def clear_memory_bank(self, model):
        """memory_bank.py:231-260: empty tensors in all five slots, mb_batch = 0."""
        target = _unwrap(model)
        if getattr(target, "mb_batch", 0):
            _info(self.logger, f"Clearing memory bank with {target.mb_batch} samples")
        target._init_memory_bank()
        return model


def _cache_features(model, loader, device, separate):
    """evaluator.py:109-171: features of every test batch.  separate=True is the multi-sentence branch (text and video
    encoded on their own)."""
    out = {k: [] for k in ("ind", "tf", "tm", "vf", "vm")}
    for batch in loader:
        text_ids, text_mask, video, video_mask, inds, _ = (t.to(device) for t in batch)
        video_mask = video_mask.view(-1, video_mask.shape[-1])
        if separate:
            tf = model.get_text_feat(text_ids, text_mask)
            vf = model.get_video_feat(video, video_mask)
        else:
            tf, vf = model.get_text_video_feat(text_ids, text_mask, video, video_mask)
        out["ind"].append(inds.reshape(-1))
        out["tf"].append(tf.float())
        out["vf"].append(vf.float())
        out["tm"].append(text_mask.view(-1, text_mask.shape[-1]))
        out["vm"].append(video_mask)
    return tuple(torch.cat(out[k], 0) for k in ("ind", "tf", "tm", "vf", "vm"))


def eval_epoch(args, model, test_dataloader, device):
    """evaluator.py:66-291 -> (text_to_video_metrics, video_to_text_metrics), the same on every rank.

    Single-sentence sets: every rank caches the features of the batches its sampler gives it, one packed all-gather + index
    scatter puts them into dataset order (:173-189), then the N x N similarity and both rank counts are computed SHARDED
    (rank r: rows [r N/W, (r+1) N/W)).  Multi-sentence sets (`dataset.multi_sentence_per_video`): every rank walks the whole
    loader, as in the reference (:114-131), keeps the video of each group's last sentence (:137-149), and the
    sentence x video matrix is again scored in row slabs (evaluator.sharded_multi_sentence_metrics)."""
    from .evaluator import dataset_order, gather_eval_features, sharded_metrics, sharded_multi_sentence_metrics
    logger = getattr(args, "logger", None)
    tracker = RetrievalMetrics(logger=logger)
    model = _unwrap(model).to(device)
    dataset = getattr(test_dataloader, "dataset", None)
    multi = bool(getattr(dataset, "multi_sentence_per_video", False))
    model.eval()
    tic = time.time()
    with torch.no_grad():
        if multi:
            cut_off_points = [p - 1 for p in dataset.cut_off_points]             # evaluator.py:98
            _info(logger, "Evaluating with multi-sentence per video setup")
            _info(logger, f"Sentences: {dataset.sentence_num}, Videos: {dataset.video_num}")
            ind, tf, tm, vf, vm = _cache_features(model, test_dataloader, device, separate=True)
            keep = torch.isin(ind, torch.tensor(cut_off_points, device=ind.device))      # evaluator.py:137-149
            vf, vm = vf[keep], vm[keep]
            toc1 = time.time()
            t2v, v2t = sharded_multi_sentence_metrics(model, tf, vf, tm.float(), vm.float(), cut_off_points, args)
        else:
            ind, tf, tm, vf, vm = _cache_features(model, test_dataloader, device, separate=False)
            if int(getattr(args, "world_size", 1)) > 1 and dist.is_initialized():
                tf, vf, tm, vm = gather_eval_features(tf, vf, ind, tm, vm, args)
            else:
                tf, vf, tm, vm = dataset_order(tf, vf, ind, tm, vm)
            toc1 = time.time()
            t2v, v2t = sharded_metrics(model, tf, vf, tm.float(), vm.float(), args)
    toc2 = time.time()
    if is_main_process() and logger is not None:
        logger.info("Evaluation timing breakdown:")
        logger.info(f"  - Feature extraction: {toc1 - tic:.2f}s")
        logger.info(f"  - Similarity + metrics: {toc2 - toc1:.2f}s")
        logger.info("=" * 80)
        logger.info("EVALUATION RESULTS")
        logger.info("=" * 80)
        tracker.log_current_metrics(t2v, v2t, (t2v["R1"] + v2t["R1"]) / 2)
    return t2v, v2t


class _Meter:
    """What trainer.py needs of metric_logger.MetricLogger when the caller passes none."""

    def __init__(self):
        self.values = {}

    def update(self, **kw):
        for k, v in kw.items():
            self.values.setdefault(k, []).append(float(v))

    def median(self, k):
        return float(np.median(self.values[k][-20:]))

    def global_avg(self, k):
        return float(np.mean(self.values[k]))


def _meter_value(meters, name, kind):
    if isinstance(meters, _Meter):
        return getattr(meters, kind)(name)
    return getattr(getattr(meters, name), kind)


def train_epoch(epoch, args, model, train_dataloader, device, n_gpu, optimizer, scheduler, global_step, max_steps,
                val_dataloader, meters=None):
    """trainer.py:18-221 -> (average loss, global_step, best text->video metrics, best video->text metrics).

    One step = forward (encoders, packed exchange, the HIP loss head, bank push) + backward + clip_grad_norm(1.0) +
    optimizer / scheduler step + logit-scale clamp (:114-119), validation every 3 * n_display steps and at step 1
    (:171-199).  The running loss stays on the device: one host sync per logged step, not one per step (:203)."""
    logger = getattr(args, "logger", None)
    tracker = RetrievalMetrics(logger=logger)
    meters = meters if meters is not None else _Meter()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    model.train()
    log_step = int(getattr(args, "n_display", 50))
    target = _unwrap(model)
    total_loss = None
    end = time.time()
    for step, batch in enumerate(train_dataloader, start=1):
        global_step += 1
        data_time = time.time() - end
        if n_gpu == 1:
            batch = tuple(t.to(device=device, non_blocking=True) for t in batch)
        text_ids, text_mask, video, video_mask, inds, idx = batch
        losses = model(text_ids, text_mask, video, video_mask, idx, global_step, logger)
        if n_gpu > 1:
            losses = tuple(v.mean() for v in losses)
        loss = losses[0]
        if getattr(args, "detect_grad", False):
            with torch.autograd.detect_anomaly():
                loss.backward()
        else:
            loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        optimizer.step()
        if scheduler is not None:
            scheduler.step()
        optimizer.zero_grad()
        torch.clamp_(target.clip.logit_scale.data, max=float(np.log(100)))          # trainer.py:114-119
        total_loss = loss.detach().clone() if total_loss is None else total_loss + loss.detach()
        batch_time = time.time() - end
        end = time.time()
        logging_now = global_step % log_step == 0 or global_step == 1
        if logging_now or not isinstance(meters, _Meter):
            from .dist import reduce_losses
            red = reduce_losses(losses, args).tolist()                               # one reduce instead of five
            meters.update(time=batch_time, data=data_time, loss=red[0], centrality_loss=red[1], uniform_loss=red[2],
                          neighbor_loss=red[3], kl_loss=red[4])
        if logging_now and is_main_process() and logger is not None:
            eta = str(timedelta(seconds=int(_meter_value(meters, "time", "global_avg") * (max_steps - global_step))))
            lr = optimizer.get_lr()[0] if hasattr(optimizer, "get_lr") else optimizer.param_groups[0]["lr"]
            logger.info(" | ".join([
                f"Epoch: {epoch}/{args.epochs}", f"Iter: {global_step}/{max_steps}",
                f"Loss: {_meter_value(meters, 'loss', 'median'):.4f}",
                f"C-Loss: {_meter_value(meters, 'centrality_loss', 'median'):.4f}",
                f"U-Loss: {_meter_value(meters, 'uniform_loss', 'median'):.4f}",
                f"N-Loss: {_meter_value(meters, 'neighbor_loss', 'median'):.4f}",
                f"KL-Loss: {_meter_value(meters, 'kl_loss', 'median'):.4f}", f"LR: {lr:.8f}",
                f"LogitScale: {float(target.clip.logit_scale.detach().exp()):.2f}", f"ETA: {eta}"]))
        if val_dataloader is not None and (global_step % (log_step * 3) == 0 or global_step == 1):
            _info(logger, "=" * 80)
            _info(logger, f"Running validation at step {global_step}")
            t2v, v2t = eval_epoch(args, model, val_dataloader, device)
            if int(getattr(args, "local_rank", 0)) == 0:
                updated, _ = tracker.update_best_metrics(t2v, v2t, t2v["R1"], v2t["R1"])
                if updated:
                    tracker.log_best_metrics()
                    save = getattr(args, "save_model_fn", None)                   # the reference imports main.save_model
                    if getattr(args, "save_model", False) and save is not None:
                        _info(logger, f"New best model saved to: {save(epoch, args, model, type_name='best')}")
            model.train()
            _info(logger, "=" * 80)
    n_steps = max(len(train_dataloader), 1)
    total = float(total_loss) / n_steps if total_loss is not None else 0.0
    _info(logger, "=" * 80)
    _info(logger, f"EPOCH {epoch} SUMMARY")
    _info(logger, f"Average loss: {total:.4f}")
    _info(logger, "=" * 80)
    best = tracker.get_best_metrics()
    return total, global_step, best["text_to_video"], best["video_to_text"]


def get_logger(name="neighborretr_amd", level=logging.INFO):
    """A plain stdout logger for callers that have none (the reference builds its own in utils/setup.py)."""
    logger = logging.getLogger(name)
    if not logger.handlers:
        h = logging.StreamHandler()
        h.setFormatter(logging.Formatter("%(asctime)s - %(levelname)s -   %(message)s", "%m/%d/%Y %H:%M:%S"))
        logger.addHandler(h)
    logger.setLevel(level)
    return logger
