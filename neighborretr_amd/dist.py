"""The path's one exchange step: gathering every rank's features / masks / ids before the
batch x batch similarity (reference: 5 x all_gather + barrier, modeling.py:274-280 via
until_module.py:367-388).

`packed_allgather` moves all five tensors in ONE collective: each rank packs
[text_feat f32 | video_feat f32 | idx i64 | text_mask u8 | video_mask u8] into one byte buffer,
one all_gather_into_tensor (RCCL over xGMI on the GPU box; gloo in the CPU tests) fills a
[W, bytes] buffer, and the pieces are sliced back out.  At b=16, Nt=24, Nv=12 a shard is ~1.2 MB:
the step is latency-bound, so one launch instead of five (plus the reference's barrier, which is
dropped: the collective already orders the data) is what matters.

Returned dtypes are the same on every device: features fp32, idx int64, masks fp32 (the multipliers
every kernel of the head reads; the reference hands int64 masks on and converts later).

Backward keeps the reference's semantics (AllGather.backward): every rank differentiates the
full replicated loss, so the gradient of the gathered features is just this rank's slice -- no
reduction.
"""
import torch
import torch.distributed as dist

from . import comm


def _world(args):
    return int(getattr(args, "world_size", 1))


class PackedAllGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, text_feat, video_feat, idx, text_mask, video_mask, args):
        W = _world(args)
        b = text_feat.shape[0]
        ctx.rank, ctx.b, ctx.W = int(getattr(args, "local_rank", 0)), b, W
        ctx.sharded = bool(getattr(args, "shard_loss", False))
        if dist.is_initialized() or comm.backend() == "emulated":
            ctx.rank = comm.get_rank()
        if W == 1:
            return text_feat.view_as(text_feat), video_feat.view_as(video_feat), idx, text_mask, video_mask
        dev = text_feat.device
        tf = text_feat.detach().float().contiguous()
        vf = video_feat.detach().float().contiguous()
        ix = idx.to(torch.int64).contiguous()
        kinds = [0, 0, 0, 0, 0]
        if dev.type == "cuda":
            from . import ops
            (tm, kinds[3]), (vm, kinds[4]) = ops.mask_piece(text_mask), ops.mask_piece(video_mask)     # converted by the pack kernel
        else:
            tm = text_mask.to(torch.uint8).contiguous()
            vm = video_mask.to(torch.uint8).contiguous()
        pieces = [tf, vf, ix, tm, vm]
        sizes = [p.numel() * (1 if k else p.element_size()) for p, k in zip(pieces, kinds)]
        offs = [sum(sizes[:k]) for k in range(5)]
        total = (sum(sizes) + 15) // 16 * 16
        send = torch.empty(total, dtype=torch.uint8, device=dev)
        recv = torch.empty(W * total, dtype=torch.uint8, device=dev)
        if dev.type == "cuda":
            # one launch packs, one collective moves, one launch unpacks -- straight into `out` when the caller
            # supplied static destinations (a captured graph reads them), the masks already as fp32 multipliers
            ops.pack_shard(pieces, send, offs, kinds)
            comm.all_gather_into_tensor(recv, send)
            out = getattr(args, "_gather_out", None)
            if out is None:
                out = (torch.empty((W * b,) + tuple(tf.shape[1:]), dtype=torch.float32, device=dev),
                       torch.empty((W * b,) + tuple(vf.shape[1:]), dtype=torch.float32, device=dev),
                       torch.empty((W * b,) + tuple(ix.shape[1:]), dtype=torch.int64, device=dev),
                       torch.empty((W * b,) + tuple(tm.shape[1:]), dtype=torch.float32, device=dev),
                       torch.empty((W * b,) + tuple(vm.shape[1:]), dtype=torch.float32, device=dev))
            ops.unpack_gathered(recv, W, total, sizes, offs, list(out), [False, False, False, True, True])
            g_tf, g_vf, g_ix, g_tm, g_vm = out
            ctx.mark_non_differentiable(g_ix, g_tm, g_vm)
            return g_tf, g_vf, g_ix, g_tm, g_vm
        parts = [p.view(-1).view(torch.uint8) for p in pieces]
        for p, n, o in zip(parts, sizes, offs):
            send[o:o + n] = p
        comm.all_gather_into_tensor(recv, send)
        recv = recv.view(W, total)

        def take(k, dtype, shape):
            o = offs[k]
            return recv[:, o:o + sizes[k]].contiguous().view(-1).view(dtype).view((W * b,) + tuple(shape))
        g_tf = take(0, torch.float32, tf.shape[1:])
        g_vf = take(1, torch.float32, vf.shape[1:])
        g_ix = take(2, torch.int64, ix.shape[1:])
        g_tm = take(3, torch.uint8, tm.shape[1:]).float()
        g_vm = take(4, torch.uint8, vm.shape[1:]).float()
        ctx.mark_non_differentiable(g_ix, g_tm, g_vm)
        return g_tf, g_vf, g_ix, g_tm, g_vm

    @staticmethod
    def backward(ctx, g_tf, g_vf, g_ix, g_tm, g_vm):
        if ctx.W == 1:
            return g_tf, g_vf, None, None, None, None
        sl = slice(ctx.b * ctx.rank, ctx.b * (ctx.rank + 1))
        if not ctx.sharded:
            return g_tf[sl], g_vf[sl], None, None, None, None
        # sharded loss (neighborretr_amd.sharded): every rank holds W x its share of dL/dX for ALL samples; the mean over
        # ranks of this rank's rows is dL/dX of its own samples (the AllGather2 pattern, until_module.py:408-412)
        outs = []
        for g in (g_tf, g_vf):
            g = g.contiguous()
            if comm.backend() == "gloo":
                g = g.clone()
                comm.all_reduce(g)
                outs.append(g[sl] / ctx.W)
            else:
                o = torch.empty((ctx.b,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
                comm.reduce_scatter_tensor(o, g)
                outs.append(o / ctx.W)
        return outs[0], outs[1], None, None, None, None


def packed_gather_raw(text_feat, video_feat, idx, text_mask, video_mask, args):
    """The exchange step up to the collective, without autograd: pack, one all-gather -> (receive buffer uint8 [W * record],
    layout dict: W, b, record, sizes, offs, shapes).  GPU tensors only.  What nr_bank_absorb_gathered and unpack_raw read."""
    from . import ops
    W = _world(args)
    dev = text_feat.device
    (tm, k_t), (vm, k_v) = ops.mask_piece(text_mask), ops.mask_piece(video_mask)        # int64 / fp32 masks: converted by the pack kernel
    kinds = [0, 0, 0, k_t, k_v]
    pieces = [text_feat.detach().float().contiguous(), video_feat.detach().float().contiguous(), idx.to(torch.int64).contiguous(), tm, vm]
    sizes = [p.numel() * (1 if k else p.element_size()) for p, k in zip(pieces, kinds)]
    offs = [sum(sizes[:k]) for k in range(5)]
    total = (sum(sizes) + 15) // 16 * 16
    send = torch.empty(total, dtype=torch.uint8, device=dev)
    recv = torch.empty(W * total, dtype=torch.uint8, device=dev)
    ops.pack_shard(pieces, send, offs, kinds)
    comm.all_gather_into_tensor(recv, send)
    return recv, dict(W=W, b=text_feat.shape[0], record=total, sizes=sizes, offs=offs, shapes=[tuple(p.shape[1:]) for p in pieces])


def unpack_raw(recv, lay):
    """The five gathered tensors (features f32, ids i64, masks f32) of a receive buffer of packed_gather_raw."""
    from . import ops
    W, b, dev = lay["W"], lay["b"], recv.device
    dt = (torch.float32, torch.float32, torch.int64, torch.float32, torch.float32)
    out = [torch.empty((W * b,) + shp, dtype=t, device=dev) for shp, t in zip(lay["shapes"], dt)]
    ops.unpack_gathered(recv, W, lay["record"], lay["sizes"], lay["offs"], out, [False, False, False, True, True])
    return tuple(out)


def packed_allgather(text_feat, video_feat, idx, text_mask, video_mask, args):
    """-> (text_feat, video_feat, idx, text_mask, video_mask) with the batch dim multiplied by W."""
    return PackedAllGather.apply(text_feat, video_feat, idx, text_mask, video_mask, args)


def reduce_losses(losses, args):
    """The trainer's five reduce(dst=0) calls for logging (setup.py:72-94) as one: rank 0 receives
    the mean over ranks of the stacked scalars."""
    W = _world(args)
    stacked = torch.stack([l.detach().float() for l in losses])
    if W < 2:
        return stacked
    dist.reduce(stacked, dst=0)
    if dist.get_rank() == 0:
        stacked /= W
    return stacked
