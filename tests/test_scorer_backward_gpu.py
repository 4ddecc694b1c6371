"""GPU: the FUSED token-scorer backward (backward._mlp_backward_hip -- what every training step runs: nr_token_mlp_bwd_hidden
recomputes the hidden layer from the normalised bf16 pairs like the forward kernel, grouped GEMMs for dX / dW1, grouped column
sums) pinned ELEMENT BY ELEMENT against autograd through Linear(512,1024)-ReLU-Linear(1024,1) (modeling.py:485-492) in fp64
on the token sets of the reference fixtures c1_b16 and c2_b128 (batch tokens + memory-bank tokens of either modality), in
both precision plans.

Two-part bar per gradient tensor, deviations relative to the tensor's largest |entry|:
    >= 99.9 % of the entries (all but 8 for the small tensors) within BULK,  every entry within 2e-2;
    BULK = 2e-3, except dW1 on the "bf16" plan: 8e-3 -- there the bank tokens' share of dW1 = dh^T X is a ONE-pass bf16 product
    (backward.ONE_PASS_WEIGHT_GRAD: their forward ran one-pass too), 2^-9 relative per term over K = 3072 / 12288 bank tokens.
The second part is there for ONE effect that is not rounding: a hidden unit whose pre-activation is within ~1e-5 of zero can
take the other ReLU branch than in fp64, which moves the entries that unit touches by up to ~1e-2 of the maximum
(DESIGN.md "Precision plan").  The count of such outliers (entries beyond 2e-3) is printed per tensor, together with the number
of (token, unit) pairs that are at risk (|pre-activation| < 2e-5 of the row's largest, non-zero upstream gradient)."""
import numpy as np
import pytest
import torch

from neighborretr_amd import backward, head, hip, modeling, ops
from util import golden, params, problem

pytestmark = pytest.mark.gpu
DEV = "cuda"
BULK, TAIL = 2e-3, 2e-2


def _upstream(w, seed):
    """A realistic upstream gradient of the logits: softmax backward of a random gradient of the token weights (zero on
    masked tokens, sums to zero over each sample)."""
    g = torch.Generator().manual_seed(seed)
    d_w = torch.randn(w.shape, generator=g).to(w.device)
    return ops.token_softmax_bwd(w, d_w.contiguous())


def _oracle(feats, dls, P, name, n_dx):
    """fp64 autograd through the scorer on the concatenated token sets (oracle arithmetic: nr_oracle.token_weights' MLP)."""
    X = torch.cat([f.reshape(-1, f.shape[-1]) for f in feats]).double().cpu().requires_grad_(True)
    dl = torch.cat([d.reshape(-1) for d in dls]).double().cpu()
    W1, b1 = P[name + ".0.weight"].double().requires_grad_(True), P[name + ".0.bias"].double().requires_grad_(True)
    W2, b2 = P[name + ".2.weight"].double().requires_grad_(True), P[name + ".2.bias"].double().requires_grad_(True)
    pre = X @ W1.t() + b1
    logits = torch.relu(pre) @ W2.t() + b2
    (logits.reshape(-1) * dl).sum().backward()
    at_risk = int(((pre.detach().abs() < 2e-5 * pre.detach().abs().amax(1, keepdim=True)) & (dl != 0)[:, None]).sum())
    return (W1.grad, b1.grad, W2.grad, b2.grad, X.grad[:n_dx]), at_risk


@pytest.mark.parametrize("name", ["c1_b16", "c2_b128"])
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_fused_scorer_backward_element_by_element(name, precision):
    g = golden(name)
    B, Nt, Nv, M, K = (int(g[k]) for k in ("B", "Nt", "Nv", "M", "K"))
    x = problem(int(g["seed"]), B, Nt, Nv, M, device=DEV)
    P = params()
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K), precision=precision)
    m.load_state_dict(P, strict=False)
    m = m.to(DEV).train()
    _, p_mlp, p_bank = head.precision_plan(m._prec())
    jobs, want = [], []
    for which, scorer, N, seed in (("text", "text_weight_fc", Nt, 1), ("video", "video_weight_fc", Nv, 2)):
        feat, mask = x[which + "_feat"], x[which + "_mask"].float()
        bfeat, bmask = x["mb_feat_" + which[0]], x["mb_mask_" + which[0]].float()
        sw = m.scorer_weights(scorer)
        sets, dls = [], []
        for f, mk, n, prec, sd in ((feat, mask, B, p_mlp, seed), (bfeat, bmask, M, p_bank, seed + 10)):
            prep = ops.prepare_tokens(f, mk, want_lo=True)
            w, _ = head.token_weights(prep, mk, sw, n, N, prec)
            dl = _upstream(w, sd)
            sets.append((prep, f.reshape(-1, f.shape[-1]), dl, prec))
            dls.append(dl)
        jobs.append(dict(sw=sw, sets=sets, add_to=None))
        want.append(_oracle([feat, bfeat], dls, P, scorer, B * N))
    got = backward._mlp_backward_hip(jobs)
    torch.cuda.synchronize()
    report, failures = [], []
    for which, mine, (ref, at_risk) in zip(("text", "video"), got, want):
        for tname, a, r in zip(("dW1", "db1", "dW2", "db2", "dX"), mine, ref):
            a, r = a.detach().double().cpu().reshape(-1), r.reshape(-1)
            scale = float(r.abs().max())
            e = (a - r).abs() / scale
            bulk = 8e-3 if (precision == "bf16" and tname == "dW1") else BULK
            outliers, worst = int((e > bulk).sum()), float(e.max())
            q = torch.quantile(e[:: max(1, e.numel() // 200000)].float(), torch.tensor([0.5, 0.999])).tolist()
            report.append(f"{which}.{tname}: median {q[0]:.1e}, 99.9 % {q[1]:.1e}, max {worst:.2e}, beyond {bulk:g}: {outliers}/{e.numel()}")
            failures += [(which, tname, "bulk", outliers, e.numel())] if outliers > max(8, 1e-3 * e.numel()) else []
            failures += [(which, tname, "tail", worst)] if worst > TAIL else []
        report.append(f"{which}: {at_risk} (token, unit) pairs at ReLU risk")
    print(f"\n[{name} {precision}] fused scorer backward vs fp64 autograd (relative to each tensor's largest entry):\n  " + "\n  ".join(report))
    assert not failures, failures
