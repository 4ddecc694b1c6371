"""Loss classes and the all-gather autograd function of the reference's until_module.py
(NeighborRetr/models/until_module.py:56-412), same class names and call signatures, computed by
the HIP row-loss / Sinkhorn kernels.

The training step does not go through these classes one by one: `NeighborRetr._compute_losses`
uses the fused path (neighborretr_amd.functional.head_losses), which evaluates all four terms for
both directions in one kernel.  The classes exist so that code written against the reference's
API (`CentralityWeightingLoss()(S, w)` ...) keeps working, each call running the same fused row
kernel with neutral inputs for the terms it does not need.
"""
import torch
from torch import nn


class _RowLossModule(nn.Module):
    def __init__(self, config=None):
        super().__init__()

    @staticmethod
    def _neutral(S):
        B = S.shape[0]
        z = torch.zeros((B, B), dtype=torch.float32, device=S.device)
        v = torch.zeros((B,), dtype=torch.float32, device=S.device)
        one = torch.ones((1,), dtype=torch.float32, device=S.device)
        return z, v, one


class CentralityWeightingLoss(_RowLossModule):
    """-mean_i w_i * log_softmax(S)[i,i]   (until_module.py:303-328); S arrives pre-scaled."""

    def forward(self, similarity_matrix, centrality_weights):
        from .functional import row_loss_terms
        S = similarity_matrix.float().contiguous()
        z, v, one = self._neutral(S)
        w = centrality_weights.float().contiguous()
        if w.dim() != 1 or w.shape[0] != S.shape[0]:
            # the reference multiplies diag_log_probs [B] by the weights and fails to broadcast (until_module.py:321)
            raise RuntimeError(f"The size of tensor a ({S.shape[0]}) must match the size of tensor b ({w.shape[-1]}) at "
                               "non-singleton dimension 1: one centrality weight per sample expected (several global "
                               "tokens per sample: see config.centrality_multi_token)")
        rl = row_loss_terms(S, z, z, z, v, v, w, w, one, 0, 1.0)
        return rl[0, 0].mean()


class NeighborAdjustingLoss(_RowLossModule):
    """until_module.py:161-211."""

    def forward(self, similarity_matrix, memory_bank_matrix, num_neighbors, temperature):
        from .functional import row_loss_terms
        S = similarity_matrix.float().contiguous()
        if num_neighbors > S.shape[0]:
            raise IndexError("num_neighbors exceeds the batch size (the reference fails the same way, "
                             "until_module.py:119-123)")
        z, v, one = self._neutral(S)
        c = memory_bank_matrix.float().sum(-1) / memory_bank_matrix.shape[-1]
        rl = row_loss_terms(S, z, z, z, c.contiguous(), c.contiguous(), v, v, one, int(num_neighbors), float(temperature))
        return rl[0, 2].mean()


class UniformRegularizationLoss(_RowLossModule):
    """Sinkhorn-target cross entropy (until_module.py:214-291)."""

    def sinkhorn_algorithm(self, scores, beta=0.3, num_iterations=50):
        from . import ops
        return ops.sinkhorn_targets(scores.detach().float().contiguous(), beta, num_iterations)[0]

    def forward(self, similarity_matrix, logit_scale, beta=0.3, num_iterations=50):
        from .functional import row_loss_terms
        G = similarity_matrix.float().contiguous()
        z, v, one = self._neutral(G)
        tgt = self.sinkhorn_algorithm(G, beta, num_iterations)
        rl = row_loss_terms(z, G, tgt, tgt, v, v, v, v, one, 0, float(logit_scale))
        return rl[0, 1].mean()


class KLDivergenceLoss(_RowLossModule):
    """kl_div(log_softmax(G), softmax(S), reduction='mean')   (until_module.py:339-359)."""

    def forward(self, global_similarity, local_similarity):
        from .functional import row_loss_terms
        G = global_similarity.float().contiguous()
        S = local_similarity.float().contiguous()
        z, v, one = self._neutral(S)
        rl = row_loss_terms(S, G, z, z, v, v, v, v, one, 0, 1.0)
        return rl[0, 3].sum() / (S.shape[0] * S.shape[1])


class AllGather(torch.autograd.Function):
    """all_gather + cat on dim 0; backward = this rank's slice of the gradient, no reduction
    (until_module.py:367-388).  One collective per tensor, written straight into the output
    buffer (all_gather_into_tensor) instead of a list of W tensors + cat."""

    @staticmethod
    def forward(ctx, tensor, args):
        ctx.rank = args.local_rank
        ctx.batch_size = tensor.shape[0]
        if args.world_size == 1:
            return tensor
        tensor = tensor.contiguous()
        out = torch.empty((args.world_size * tensor.shape[0],) + tuple(tensor.shape[1:]), dtype=tensor.dtype,
                          device=tensor.device)
        torch.distributed.all_gather_into_tensor(out, tensor)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output[ctx.batch_size * ctx.rank: ctx.batch_size * (ctx.rank + 1)], None


class AllGather2(torch.autograd.Function):
    """Variant whose backward sums the gradient over ranks before slicing (until_module.py:391-412):
    the correct adjoint when every rank evaluates only its own share of the loss."""

    @staticmethod
    def forward(ctx, tensor, args):
        ctx.rank = args.local_rank
        ctx.batch_size = tensor.shape[0]
        ctx.world_size = args.world_size
        if args.world_size == 1:
            return tensor
        tensor = tensor.contiguous()
        out = torch.empty((args.world_size * tensor.shape[0],) + tuple(tensor.shape[1:]), dtype=tensor.dtype,
                          device=tensor.device)
        torch.distributed.all_gather_into_tensor(out, tensor)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        if ctx.world_size == 1:
            return grad_output, None
        g = grad_output.contiguous()
        if torch.distributed.get_backend() == "gloo":          # gloo has no reduce_scatter: the
            g = g.clone()                                       # reference's all_reduce + slice
            torch.distributed.all_reduce(g)
            return g[ctx.rank * ctx.batch_size:(ctx.rank + 1) * ctx.batch_size], None
        out = torch.empty((ctx.batch_size,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        torch.distributed.reduce_scatter_tensor(out, g)
        return out, None
