"""CPU: host-side logic that needs no GPU -- the token-clustering stage against the oracle, the
C-ABI surface of the built library, loud failure of the product path without a device."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import nr_oracle as O
from neighborretr_amd import hip, modeling, synth
from util import maxdiff, noise, params, problem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(P):
    m = modeling.NeighborRetr(modeling.default_config())
    missing, unexpected = m.load_state_dict(P, strict=False)
    assert not unexpected
    assert all(k.startswith("clip.") for k in missing), missing
    return m


@pytest.mark.parametrize("B,Nt,Nv,blank", [(16, 24, 12, None), (8, 64, 64, None), (6, 20, 9, None)])
def test_merge_global_features_matches_oracle(B, Nt, Nv, blank):
    x = problem(77, B, Nt, Nv, 4)
    P = params()
    nz = noise(77, B, Nt, Nv)
    m = _model(P)
    gt, gv = m.merge_global_features(x["text_feat"], x["video_feat"], x["text_mask"], x["video_mask"], nz)
    gt_o, gv_o = O.merge_global_features(x["text_feat"], x["video_feat"], x["text_mask"], x["video_mask"], P, nz)
    assert gt.shape == gt_o.shape and gv.shape == gv_o.shape
    assert maxdiff(gt, gt_o) < 1e-5 and maxdiff(gv, gv_o) < 1e-5


def test_state_dict_names_cover_reference_head():
    m = modeling.NeighborRetr(modeling.default_config())
    have = set(m.state_dict().keys())
    want = set(synth.head_param_shapes().keys())
    assert want <= have, sorted(want - have)
    for k, shp in synth.head_param_shapes().items():
        assert tuple(m.state_dict()[k].shape) == tuple(shp), k
    assert "clip.logit_scale" in have


def test_library_exports_every_declared_symbol():
    assert os.path.exists(hip.LIB_PATH), "build the extension first (python -m neighborretr_amd.build)"
    header = open(os.path.join(ROOT, "include", "nr_hip.h")).read()
    declared = set(re.findall(r"^(?:int|size_t)\s+(nr_\w+)\s*\(", header, flags=re.M))
    assert len(declared) >= 20
    lib = ctypes.CDLL(hip.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(hip.exported_symbols()), declared ^ set(hip.exported_symbols())
    assert lib.nr_version() == 5
    assert lib.nr_prepare_parts(10) == 1 and lib.nr_prepare_parts(3072) == 192 and lib.nr_prepare_parts(10 ** 6) == 256


def test_descriptor_structs_have_the_layout_the_library_was_built_with():
    """Every descriptor struct of include/nr_hip.h that hip.py mirrors in ctypes: same size as the compiled one (nr_struct_size),
    and the sizes the header's comments state."""
    lib = hip.lib()                                        # load-time check: raises on any mismatch
    header = open(os.path.join(ROOT, "include", "nr_hip.h")).read()
    typedefs = set(re.findall(r"^\}\s*(Nr\w+);", header, flags=re.M))
    assert typedefs == set(hip.STRUCTS), typedefs ^ set(hip.STRUCTS)
    for name, cls in hip.STRUCTS.items():
        assert int(lib.nr_struct_size(name.encode())) == ctypes.sizeof(cls) > 0, name
        stated = re.search(r"\}\s*" + name + r";\s*/\*\s*(\d+) bytes", header)
        if stated:
            assert int(stated.group(1)) == ctypes.sizeof(cls), name
    assert int(lib.nr_struct_size(b"NrNoSuchStruct")) == 0


def test_similarity_backward_planning_no_gpu_needed():
    """Host-only parts of the grouped similarity backward: which token counts the matrix-core kernel takes (blocks of at most 32
    sample pairs), and the workspace planner on the step's four products (no device call: 256 CUs assumed)."""
    lib = hip.lib()
    ok = {(24, 12), (12, 24), (24, 24), (16, 24), (24, 16)}
    for nt in (4, 8, 12, 16, 24, 32, 20):
        for nv in (4, 8, 12, 16, 24):
            assert bool(lib.nr_local_level_bwd_mfma_supported(nt, nv, 512)) == ((nt, nv) in ok), (nt, nv)
    assert not lib.nr_local_level_bwd_mfma_supported(24, 12, 320)            # d % 256
    B, M, Nt, Nv, d = 128, 512, 24, 12, 512
    items = (hip.SimBwdItem * 4)()
    dummy = ctypes.create_string_buffer(64)
    ptr = ctypes.addressof(dummy)
    for it, (side, A, Bv) in zip(items, ((0, B, B), (0, B, M), (1, B, B), (1, M, B))):
        for f in ("dS", "oT_hi", "w_self", "w_other", "arg_v", "arg_t", "d_x"):
            setattr(it, f, ptr)
        it.side, it.A, it.Nt, it.Bv, it.Nv, it.d, it.ds_scale = side, A, Nt, Bv, Nv, d, 1.0
        n_other = (Bv * Nv) if side == 0 else (A * Nt)
        it.ldk = (n_other + 95) // 96 * 96
    nbytes = int(lib.nr_local_level_bwd_group_workspace_bytes(4, items))
    text, video = B * Nt * d * 4, B * Nv * d * 4
    assert nbytes > 2 * (text + video)                     # at least one slab per product
    assert nbytes <= 32 * 2 * (text + video) + 256         # and never more than 32 chunks each
    items[1].ldk -= 96                                      # operand that does not cover the product's tokens: refused
    assert int(lib.nr_local_level_bwd_group_workspace_bytes(4, items)) == 0


def test_round3_entry_points_refuse_bad_arguments_before_any_launch():
    """Argument checks of the grouped backward entry points return NR_EINVAL / NR_EUNSUPPORTED on the host (no GPU needed)."""
    lib = hip.lib()
    EINVAL, EUNSUP = -1, -2
    buf = ctypes.create_string_buffer(4096)
    ptr = ctypes.addressof(buf)
    assert lib.nr_local_level_bwd_group(0, None, None, 0, None) == EINVAL
    assert lib.nr_local_level_bwd_group(5, (hip.SimBwdItem * 5)(), ptr, 4096, None) == EINVAL        # > 4 products
    assert lib.nr_pool_weight_bwd_group(0, None, None) == EINVAL
    job = (hip.PoolWJob * 1)()
    job[0].d_w, job[0].n_src, job[0].N = ptr, 3, 12                                                   # > 2 sources
    assert lib.nr_pool_weight_bwd_group(1, job, None) == EINVAL
    ss = (hip.SlabSum * 1)()
    ss[0].out, ss[0].n, ss[0].n_src = ptr, 6, 1                                                       # n % 4 != 0
    ss[0].part[0], ss[0].n_slabs[0] = ptr, 1
    assert lib.nr_slab_sum_group(1, ss, None) == EINVAL
    op = (hip.SimBwdOperand * 1)()
    op[0].hi, op[0].out_hi, op[0].n_tok, op[0].d = ptr, ptr, 96, 96                                   # d % 64 != 0
    assert lib.nr_sim_bwd_operand_group(1, op, None) == EINVAL
    it = (hip.SplitItem * 1)()
    it[0].src, it[0].hi, it[0].rows, it[0].cols, it[0].ld = ptr, ptr, 8, 8, 8
    it[0].mode = 7
    assert lib.nr_split_group(1, it, None) == EINVAL                                                  # unknown mode
    it[0].mode, it[0].group = 3, 0
    assert lib.nr_split_group(1, it, None) == EINVAL                                                  # mode 3 without a sample size
    lp = (hip.LinearProblem * 2)()
    for k in range(2):
        lp[k].x_hi, lp[k].w_hi, lp[k].out, lp[k].M, lp[k].N, lp[k].K = ptr, ptr, ptr, 64, 64, 64
    lp[0].x_lo, lp[0].w_lo = ptr, ptr                                                                 # one problem split-bf16, one one-pass
    assert lib.nr_linear_group(2, lp, None) == EINVAL
    lp[1].x_lo, lp[1].w_lo = ptr, ptr
    lp[1].ld = 32                                                                                     # row pitch below K
    assert lib.nr_linear_group(2, lp, None) == EINVAL
    assert lib.nr_rowloss_coef(ptr, None, None, None, None, 1.0, 1.0, 1.0, 0, ptr, None) == EINVAL    # B = 0
    assert lib.nr_token_mlp_bwd_hidden(ptr, ptr, ptr, 128, 512, ptr, ptr, ptr, ptr, 1024, hip.PREC_BF16X3, ptr, ptr, ptr, 256, 4, None,
                                       None, ptr, ptr, ptr, None) == EUNSUP                          # t0 not a multiple of 8
    assert lib.nr_token_mlp_bwd_part_rows(12288, 1024, hip.PREC_BF16, 1) == 2 * 64                    # 192-row blocks
    assert lib.nr_token_mlp_bwd_part_rows(12288, 1024, hip.PREC_BF16, 0) == 2 * 96                    # 128-row blocks
    assert lib.nr_token_mlp_bwd_part_rows(3072, 1024, hip.PREC_BF16X3, 0) == 2 * 24


def test_round4_entry_points_refuse_bad_arguments_before_any_launch():
    """nr_pack_shard_convert / nr_copy_group / nr_bank_absorb_gathered: argument checks on the host; the ticket words the absorb
    launch asks for; the mask kinds the host side hands to the pack launch; bench.py knows its round-4 switches (no GPU needed)."""
    import subprocess
    import sys
    lib = hip.lib()
    EINVAL = -1
    buf = ctypes.create_string_buffer(4096)
    ptr = ctypes.addressof(buf)
    one = (ctypes.c_void_p * 1)(ptr)
    sz = (ctypes.c_size_t * 1)(16)
    off = (ctypes.c_size_t * 1)(0)
    assert lib.nr_pack_shard_convert(1, one, sz, off, (ctypes.c_int * 1)(3), ptr, None) == EINVAL        # unknown kind
    assert lib.nr_pack_shard_convert(9, one, sz, off, None, ptr, None) == EINVAL                          # > 8 pieces
    assert lib.nr_copy_group(13, one, one, sz, None) == EINVAL                                            # > 12 copies
    assert lib.nr_copy_group(1, (ctypes.c_void_p * 1)(None), one, sz, None) == EINVAL                     # null source
    assert lib.nr_bank_absorb_gathered(None, None) == EINVAL
    assert lib.nr_token_weights_fwd_pair(None, None, hip.PREC_BF16, None) == EINVAL
    tw = hip.TokenWeightsProblem()                                                                        # all-null problem
    assert lib.nr_token_weights_fwd_pair(ctypes.byref(tw), ctypes.byref(tw), hip.PREC_BF16, None) == EINVAL
    assert lib.nr_bank_absorb_counter_words() == 16 * (1 + 2048 // 32)        # the launch's word + one per group of 32 workgroups, 64 B apart
    from neighborretr_amd import ops
    assert ops.mask_piece(torch.ones(2, 3, dtype=torch.int64))[1] == 1 and ops.mask_piece(torch.ones(2, 3))[1] == 2
    m8, k8 = ops.mask_piece(torch.ones(2, 3, dtype=torch.bool))
    assert k8 == 0 and m8.dtype == torch.uint8
    import neighborretr_amd.interleave  # noqa: F401  (importable without a GPU)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and all(f in r.stdout for f in ("--no_overlap", "--overlap_tries", "--sync_step", "--unroll"))


def test_k_slicing_of_a_weight_gradient():
    from neighborretr_amd.backward import _apportion
    assert _apportion([3072, 1536], 8) == [5, 3]
    assert _apportion([12288, 6144], 8) == [5, 3]
    assert _apportion([100, 1, 1], 8) == [6, 1, 1]
    assert _apportion([1, 1, 1], 2) == [1, 1, 1]            # never fewer than one problem per set
    assert sum(_apportion([7, 5, 3, 2], 8)) == 8


def test_tile_query_no_gpu_needed():
    assert hip.local_level_tiles(128, 24, 128, 12) == (32, 16)      # 4 texts x 8 videos per 96x96 block
    assert hip.local_level_tiles(128, 64, 1024, 64) == (32, 256)    # one pass: 4 x 4 per 256x256 block on 8 waves (two texts per wave strip)
    assert hip.local_level_tiles(130, 64, 1024, 64) == (65, 256)    # ... 2 x 4 per 128x256 block when the texts do not come in fours
    assert hip.local_level_tiles(128, 64, 1024, 64, hip.PREC_BF16X3) == (32, 256)    # split-bf16: the same blocks, three accumulated passes
    assert hip.local_level_tiles(6, 64, 10, 64, hip.PREC_BF16X3) == (3, 5)           # ... too few of them: 2 x 2 per 128x128 split block
    assert hip.local_level_tiles(128, 24, 512, 12, hip.PREC_BF16) == (16, 16)      # bank product: 8 x 32 per 192x384 block
    assert hip.local_level_tiles(128, 24, 512, 12, hip.PREC_BF16X3) == (16, 16)    # split-bf16: the same blocks, three accumulated passes
    with pytest.raises(hip.NrHipError):
        hip.local_level_tiles(4, 200, 4, 12)                         # > 128 tokens per sample: unsupported


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu():
    from neighborretr_amd import ops
    with pytest.raises(hip.NrHipError):
        ops.prepare_tokens(torch.randn(4, 3, 512))
    m = modeling.NeighborRetr(modeling.default_config())
    x = problem(1, 4, 24, 12, 8)
    with pytest.raises(hip.NrHipError):
        with torch.no_grad():
            m.local_level(x["text_feat"], x["video_feat"], x["text_mask"], x["video_mask"])


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "neighborretr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "nr_oracle" not in src and "import oracle" not in src, f


def test_multi_sentence_metrics_against_counting():
    """utils/metrics.py:82-148 semantics by brute force: sim[i, s, j] = sentence s of video i vs video j."""
    from neighborretr_amd.metrics import RetrievalMetrics
    g = torch.Generator().manual_seed(1)
    nv, ms = 20, 3
    sim = torch.randn(nv, ms, nv, generator=g)
    for i in range(nv):
        sim[i, :, i] += 1.5
        if i % 3 == 0:
            sim[i, 2, :] = float("-inf")                    # video i has only two sentences
        if i % 5 == 0:
            sim[i, 1, :] = float("nan")
    ranks = []
    for s_ in range(ms):
        for i in range(nv):
            if torch.isfinite(sim[i, s_, i]):
                ranks.append(int((sim[i, s_] > sim[i, s_, i]).sum()))
    ranks = np.array(ranks)
    res = RetrievalMetrics.tensor_text_to_video_metrics(sim.clone())
    for k in (1, 5, 10, 50):
        assert abs(res[f"R{k}"] - 100.0 * np.mean(ranks < k)) < 1e-4
    assert res["MeanR"] == pytest.approx(np.mean(ranks + 1)) and res["MR"] == float(np.sort(ranks + 1)[(len(ranks) - 1) // 2])
    v2t = RetrievalMetrics.tensor_video_to_text_sim(sim.clone())
    ref = torch.where(torch.isnan(sim), torch.full_like(sim, float("-inf")), sim).max(1)[0].t()
    assert v2t.shape == (nv, nv) and torch.equal(v2t, ref)


def test_fresh_tcblock_initialises_like_the_reference():
    """cluster.py:918-932: a new TCBlock applies trunc_normal(0.02) to q / kv / proj, zero biases, LayerNorm 1 / 0 --
    also when built after the model-wide init (modeling.py:63-72), so a run without --init_model starts from the
    reference's distribution."""
    torch.manual_seed(0)
    m = modeling.NeighborRetr(modeling.default_config())
    for name in ("text_block0", "text_block1", "video_block0", "video_block1"):
        blk = getattr(m, name)
        for lin in (blk.attn.q, blk.attn.kv, blk.attn.proj):
            w = lin.weight.detach()
            assert abs(float(w.std()) - 0.02) < 1.5e-3 and abs(float(w.mean())) < 1e-3, (name, float(w.std()))
            assert float(w.abs().max()) <= 2.0 and float(lin.bias.abs().max()) == 0.0
        assert float((blk.norm1.weight - 1).abs().max()) == 0.0 and float(blk.norm1.bias.abs().max()) == 0.0
    # the CTM modules keep torch's defaults, as in the reference (no init hook there: cluster.py:670-688)
    assert float(m.text_ctm0.score.bias.abs().max()) >= 0.0


def test_reference_import_paths_resolve_to_this_package():
    """with this repository ahead of the reference on PYTHONPATH, the reference's own import lines
    (main.py:44, training/trainer.py, training/evaluator.py:14-16, utils/memory_bank.py:17) bind the HIP-backed modules."""
    import importlib
    import neighborretr_amd.cluster as C
    import neighborretr_amd.metrics as Mx
    import neighborretr_amd.modeling as Mo
    import neighborretr_amd.training as T
    import neighborretr_amd.until_module as U
    expect = {
        "NeighborRetr.models.modeling": {"NeighborRetr": Mo.NeighborRetr, "AllGather": U.AllGather},
        "NeighborRetr.models.until_module": {"CentralityWeightingLoss": U.CentralityWeightingLoss, "AllGather": U.AllGather},
        "NeighborRetr.models.cluster": {"CTM": C.CTM, "TCBlock": C.TCBlock},
        "NeighborRetr.utils.metrics": {"RetrievalMetrics": Mx.RetrievalMetrics},
        "NeighborRetr.utils.memory_bank": {"MemoryBankManager": T.MemoryBankManager},
        "NeighborRetr.utils.comm": {"is_main_process": T.is_main_process},
        "NeighborRetr.training.trainer": {"train_epoch": T.train_epoch},
        "NeighborRetr.training.evaluator": {"eval_epoch": T.eval_epoch},
    }
    for mod, names in expect.items():
        m = importlib.import_module(mod)
        for name, obj in names.items():
            assert getattr(m, name) is obj, (mod, name)


def test_metrics_tracker_follows_the_reference_update_rule():
    """utils/metrics.py:168-203: a tie with the best R@1 updates too; the mean of the BEST t2v / v2t is tracked."""
    from neighborretr_amd.metrics import RetrievalMetrics
    tr = RetrievalMetrics()
    a = {"R1": 10.0, "R5": 30.0, "R10": 40.0, "MR": 12.0, "MeanR": 20.0}
    b = {"R1": 20.0, "R5": 35.0, "R10": 45.0, "MR": 10.0, "MeanR": 18.0}
    updated, mean = tr.update_best_metrics(a, b, a["R1"], b["R1"])
    assert updated and mean == 15.0 and tr.get_best_metrics()["score"] == 15.0
    updated, mean = tr.update_best_metrics({**a, "R1": 12.0}, {**b, "R1": 5.0}, 12.0, 5.0)
    assert updated and mean == 8.5
    best = tr.get_best_metrics()
    assert best["t2v_r1"] == 12.0 and best["v2t_r1"] == 20.0 and best["score"] == 16.0
    updated, _ = tr.update_best_metrics({**a, "R1": 1.0}, {**b, "R1": 1.0}, 1.0, 1.0)
    assert not updated
    updated, _ = tr.update_best_metrics({**a, "R1": 12.0}, {**b, "R1": 1.0})          # R@1 taken from the dictionaries
    assert updated


@pytest.mark.parametrize("B,N,C,ratio,masked", [(5, 24, 64, 1 / 6, True), (4, 12, 64, 0.25, True), (6, 4, 64, 0.25, False),
                                                (3, 3, 64, 1 / 3, False)])
def test_hand_derived_stage_backward_equals_autograd(B, N, C, ratio, masked):
    """cluster_backward.stage_backward (what the training step's fused clustering uses) against autograd through the traced
    stage of cluster.py, fp64, cluster ids fixed: inputs and all 13 parameter gradients."""
    import math
    from neighborretr_amd import cluster, cluster_backward as CB
    torch.manual_seed(B * 100 + N)
    ctm, blk = cluster.CTM(ratio, C, C, k=3).double(), cluster.TCBlock(C, 8).double()
    for p in list(ctm.parameters()) + list(blk.parameters()):
        p.data.add_(0.05 * torch.randn_like(p))
    x0 = torch.randn(B, N, C, dtype=torch.float64)
    mask = None
    if masked:
        ln = torch.randint(1, N + 1, (B,))
        ln[0], ln[1] = N, 1                                       # a full sample and one with a single valid token
        mask = (torch.arange(N)[None] < ln[:, None]).double()
    c = max(math.ceil(N * ratio), 1)
    with torch.no_grad():
        assign = cluster.dpc_knn_assign(ctm.norm(ctm.conv(x0)), c, ctm.k, mask, torch.rand(B, N, dtype=torch.float64))
    xg = x0.clone().requires_grad_(True)
    out = blk(ctm({"x": xg, "mask": mask}, assign=assign))["x"]
    g = torch.randn_like(out)
    ps = list(ctm.parameters()) + list(blk.parameters())
    ref = torch.autograd.grad(out, [xg] + ps, g)
    dx, grads = CB.stage_backward(ctm, blk, CB.saved_from_modules(ctm, blk, x0, mask, assign), g)
    assert len(grads) == len(ps) == 13
    assert float((dx - ref[0]).abs().max()) < 1e-8 * float(ref[0].abs().max())
    for p, r in zip(ps, ref[1:]):
        assert grads[p].shape == r.shape
        assert float((grads[p] - r).abs().max()) < 1e-8 * max(float(r.abs().max()), 1e-12)


def test_capture_guard_refuses_the_two_crashing_shapes_and_passes_the_shipped_ones():
    """the stream-topology rules of HIP-graph capture on ROCm 7.2 (profiles/r02_capture_refork.txt,
    r02_capture_nest.txt) live in code.  The pure bookkeeping runs on integer handles -- no GPU, and above all no execution
    of the crashing captures themselves."""
    from neighborretr_amd.capture_guard import CaptureTopologyError, StreamTopology
    ORIGIN, S1, S2, S3 = 0, 1, 2, 3
    # shape 1 (tools/capture_refork.py "refork"): s2 joined into s1, then s2 waits on s1 again
    t = StreamTopology(ORIGIN)
    t.wait(S1, ORIGIN), t.wait(S2, ORIGIN)
    t.wait(S1, S2)
    with pytest.raises(CaptureTopologyError, match="already joined"):
        t.wait(S2, S1)
    # ... the same work on a FRESH third stream is fine ("fresh")
    t = StreamTopology(ORIGIN)
    t.wait(S1, ORIGIN), t.wait(S2, ORIGIN), t.wait(S1, S2), t.wait(S3, S1)
    t.wait(ORIGIN, S1), t.wait(ORIGIN, S3)
    # shape 2 (tools/capture_nest.py): a forked stream joins a stream forked from itself
    t = StreamTopology(ORIGIN)
    t.wait(S1, ORIGIN)
    t.wait(S2, S1)
    with pytest.raises(CaptureTopologyError, match="forked from itself"):
        t.wait(S1, S2)
    # ... the same kid joined straight into the origin is fine, and so are many siblings
    t = StreamTopology(ORIGIN)
    t.wait(S1, ORIGIN), t.wait(S2, S1), t.wait(ORIGIN, S2), t.wait(ORIGIN, S1)
    t = StreamTopology(ORIGIN)
    for s_ in range(1, 17):
        t.wait(s_, ORIGIN)
    for s_ in range(1, 17):
        t.wait(ORIGIN, s_)
    # the shipped loss-only step (head.head_forward, split tail): local | side, side2, push | joins into the origin
    LOCAL, SIDE, SIDE2, PUSH = 10, 11, 12, 13
    t = StreamTopology(ORIGIN)
    t.wait(LOCAL, ORIGIN)
    t.wait(SIDE, LOCAL), t.wait(SIDE2, LOCAL), t.wait(SIDE, SIDE2), t.wait(PUSH, SIDE), t.wait(SIDE, ORIGIN)
    t.wait(ORIGIN, SIDE), t.wait(ORIGIN, PUSH)
    # round 2's attempted edit -- the push back on side2 after side.wait_stream(side2) -- is refused
    t = StreamTopology(ORIGIN)
    t.wait(LOCAL, ORIGIN), t.wait(SIDE, LOCAL), t.wait(SIDE2, LOCAL), t.wait(SIDE, SIDE2)
    with pytest.raises(CaptureTopologyError):
        t.wait(SIDE2, SIDE)
    # autograd re-enters a side stream that was joined into the ORIGIN (every captured training step does): allowed
    t = StreamTopology(ORIGIN)
    t.wait(S1, ORIGIN), t.wait(ORIGIN, S1), t.wait(S1, ORIGIN), t.wait(ORIGIN, S1)


def test_cooperative_sinkhorn_gate_is_a_pure_host_decision():
    """the one-launch Sinkhorn of 128 < B <= 1024 needs the B/32 workgroups of a direction resident together; the host
    gate (nr_sinkhorn_cooperative_gate: blocks per CU x CUs of one XCD >= B/32, and both directions fit the chip) decides
    between it and the multi-launch form.  No GPU involved."""
    gate = hip.lib().nr_sinkhorn_cooperative_gate
    assert gate(1024, 1, 256, 8) == 1 and gate(512, 1, 256, 8) == 1 and gate(192, 1, 256, 8) == 1      # MI355X, SPX: 32 CUs per XCD
    assert gate(1024, 1, 128, 8) == 0          # half the CUs (a CU mask / a smaller part): 16 per XCD < 32 workgroups
    assert gate(512, 1, 128, 8) == 1 and gate(1024, 2, 128, 8) == 1
    assert gate(1024, 1, 32, 1) == 0 and gate(512, 1, 32, 1) == 1                                       # one XCD (CPX partition): both directions share it
    assert gate(1024, 0, 256, 8) == 0          # the kernel does not fit a CU at all
    assert gate(128, 1, 256, 8) == 0 and gate(1088, 4, 256, 8) == 0 and gate(200, 1, 256, 8) == 0      # sizes the form does not cover
