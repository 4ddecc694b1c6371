"""ctypes binding of libnr_hip.so (include/nr_hip.h) for torch tensors on ROCm.

This is the only bridge between the Python host code and the HIP kernels.  There is no CPU or
eager fallback behind it: if the shared library is missing or a tensor is not a contiguous
device tensor of the right dtype, the call raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# NR_HIP_LIB: developer hook to A/B two builds of the library in one GPU session (tools/); never a fallback
LIB_PATH = os.environ.get("NR_HIP_LIB") or os.path.join(_HERE, "libnr_hip.so")

PREC_BF16 = 0
PREC_BF16X3 = 1
OUT_FULL, OUT_ROWSUM, OUT_COLSUM = 0, 1, 2
NR_EINVAL, NR_EUNSUPPORTED = -1, -2          # status codes of include/nr_hip.h
ABI_VERSION = 5                              # NR_ABI_VERSION this binding was written for (checked at load)

_lib = None

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_Z = ctypes.c_size_t

class CtmStageDesc(ctypes.Structure):
    """NrCtmStageDesc of include/nr_hip.h (field order and types are the ABI)."""
    _fields_ = ([(n, ctypes.c_int32) for n in ("n_samples", "N", "C", "k", "cnum", "heads")]
                + [("eps_ctm", _F), ("eps_n1", _F)]
                + [(n, _P) for n in ("x", "mask", "noise", "wconv_hi", "wconv_lo", "conv_bias", "ln_w", "ln_b", "sc_w", "sc_b",
                                     "n1_w", "n1_b", "wq_hi", "wq_lo", "q_bias", "wkv_hi", "wkv_lo", "kv_bias", "wp_hi", "wp_lo",
                                     "proj_bias", "workspace", "out", "assign", "x_hi", "x_lo", "out_hi", "out_lo")])


class LocalLevelProblem(ctypes.Structure):
    """NrLocalLevelProblem of include/nr_hip.h."""
    _fields_ = ([(n, _P) for n in ("t_hi", "t_lo", "v_hi", "v_lo", "w_t", "w_v", "out")]
                + [(n, ctypes.c_int) for n in ("A", "Nt", "Bv", "Nv", "d", "prec", "out_mode")])


class SplitItem(ctypes.Structure):
    """NrSplitItem of include/nr_hip.h."""
    _fields_ = ([("src", _P), ("src2", _P), ("hi", _P), ("lo", _P)]
                + [(n, ctypes.c_int32) for n in ("rows", "cols", "mode", "ld", "group", "pad_")])


class ColsumItem(ctypes.Structure):
    """NrColsumItem of include/nr_hip.h."""
    _fields_ = [("src", _P), ("dst", _P), ("rows", ctypes.c_int32), ("cols", ctypes.c_int32), ("scale", _F), ("pad_", ctypes.c_int32)]


class LinearProblem(ctypes.Structure):
    """NrLinearProblem of include/nr_hip.h."""
    _fields_ = ([(n, _P) for n in ("x_hi", "x_lo", "w_hi", "w_lo", "bias", "residual", "out")]
                + [(n, ctypes.c_int32) for n in ("M", "N", "K", "ld", "conv_n", "pad_")])


class CtmAttnBwdDesc(ctypes.Structure):
    """NrCtmAttnBwdDesc of include/nr_hip.h."""
    _fields_ = ([(n, ctypes.c_int32) for n in ("n_samples", "N", "C", "cnum", "heads")]
                + [(n, _P) for n in ("q", "kv", "score", "d_att", "d_q", "d_kv", "d_score", "dq_hi", "dq_lo", "dkv_hi", "dkv_lo")])


class CtmMidBwdDesc(ctypes.Structure):
    """NrCtmMidBwdDesc of include/nr_hip.h."""
    _fields_ = ([(n, ctypes.c_int32) for n in ("n_samples", "N", "C", "cnum")] + [("eps_ctm", _F), ("eps_n1", _F)]
                + [(n, _P) for n in ("d_qn", "d_kvn", "g", "merged_pb", "proj_b", "xn", "y", "tokw", "d_score", "mask", "n1_w",
                                     "ln_w", "sc_w", "assign", "d_y", "dy_hi", "dy_lo", "partial")])


class SimBwdItem(ctypes.Structure):
    """NrSimBwdItem of include/nr_hip.h."""
    _fields_ = ([(n, _P) for n in ("dS", "oT_hi", "oT_lo", "w_self", "w_other", "arg_v", "arg_t", "d_x")] + [("ds_scale", _F)]
                + [(n, ctypes.c_int32) for n in ("side", "ds_mode", "ldk", "A", "Nt", "Bv", "Nv", "d", "accumulate")])


class SimBwdOperand(ctypes.Structure):
    """NrSimBwdOperand of include/nr_hip.h."""
    _fields_ = [(n, _P) for n in ("hi", "lo", "out_hi", "out_lo")] + [("n_tok", ctypes.c_int32), ("d", ctypes.c_int32)]


class SlabSum(ctypes.Structure):
    """NrSlabSum of include/nr_hip.h."""
    _fields_ = [("out", _P), ("n", ctypes.c_uint64), ("accumulate", ctypes.c_int32), ("n_src", ctypes.c_int32),
                ("part", _P * 4), ("n_slabs", ctypes.c_int32 * 4)]


class PoolWSrc(ctypes.Structure):
    """NrPoolWSrc of include/nr_hip.h."""
    _fields_ = [("dS", _P), ("pool", _P), ("ds_scale", _F)] + [(n, ctypes.c_int32) for n in ("ds_mode", "A", "Bv")]


class PoolWJob(ctypes.Structure):
    """NrPoolWJob of include/nr_hip.h."""
    _fields_ = [("src", PoolWSrc * 2), ("d_w", _P)] + [(n, ctypes.c_int32) for n in ("n_src", "side", "N", "accumulate")]


class BankAbsorbDesc(ctypes.Structure):
    """NrBankAbsorbDesc of include/nr_hip.h."""
    _fields_ = ([("gathered", _P)] + [(n, ctypes.c_uint64) for n in ("record_bytes", "off_text", "off_video", "off_index", "off_text_mask",
                                                                      "off_video_mask")]
                + [(n, ctypes.c_int32) for n in ("world", "per_rank", "Nt", "Nv", "d", "capacity")]
                + [(n, _P) for n in ("bank_text", "bank_video", "bank_text_mask", "bank_video_mask", "bank_index", "shadow_text_hi",
                                     "shadow_text_lo", "shadow_video_hi", "shadow_video_lo", "shadow_text_norm", "shadow_video_norm",
                                     "ring_head", "rng_state", "counter")])


class TokenWeightsProblem(ctypes.Structure):
    """NrTokenWeightsProblem of include/nr_hip.h."""
    _fields_ = ([(n, _P) for n in ("tok_hi", "tok_lo", "norm", "w1_hi", "w1_lo", "b1", "w2", "b2", "mask", "logit_part", "counters", "w", "logits")]
                + [(n, ctypes.c_int32) for n in ("n_samples", "N", "d", "H", "n_counters", "reserved")])


STRUCTS = {"NrTokenWeightsProblem": TokenWeightsProblem, "NrBankAbsorbDesc": BankAbsorbDesc, "NrCtmStageDesc": CtmStageDesc, "NrLocalLevelProblem": LocalLevelProblem, "NrSplitItem": SplitItem,
           "NrColsumItem": ColsumItem, "NrLinearProblem": LinearProblem, "NrCtmAttnBwdDesc": CtmAttnBwdDesc,
           "NrCtmMidBwdDesc": CtmMidBwdDesc, "NrSimBwdItem": SimBwdItem, "NrSimBwdOperand": SimBwdOperand, "NrSlabSum": SlabSum,
           "NrPoolWSrc": PoolWSrc, "NrPoolWJob": PoolWJob}
SPLIT_MAX, COLSUM_MAX, LINEAR_GROUP_MAX = 48, 16, 8
SIM_BWD_GROUP_MAX, POOLW_GROUP_MAX = 4, 8
LOCAL_LEVEL_GROUP_MAX = 4
CTM_MAX_GROUP = 4
CTM_STAGE_LAUNCHES = 7

_SIGNATURES = {
    "nr_version": ([], _I),
    "nr_struct_size": ([ctypes.c_char_p], _Z),
    "nr_stream_capture_id": ([_P, ctypes.POINTER(ctypes.c_ulonglong)], _I),
    "nr_stream_create": ([ctypes.POINTER(_P)], _I),
    "nr_stream_destroy": ([_P], _I),
    "nr_prepare_parts": ([_I], _I),
    "nr_prepare_tokens": ([_P, _P, _I, _I, _I, _P, _P, _P, _P, _P], _I),
    "nr_prepare_tokens_pair": ([_P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _P], _I),
    "nr_split_bf16": ([_P, _Z, _P, _P, _P], _I),
    "nr_token_logits_fwd": ([_P, _P, _P, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P], _I),
    "nr_token_softmax": ([_P, _I, _P, _P, _I, _I, _P, _P, _P], _I),
    "nr_token_mlp_bwd_row_tiles": ([_I], _I),
    "nr_token_mlp_bwd_part_rows": ([_I, _I, _I, _I], _I),
    "nr_token_mlp_bwd_hidden": ([_P, _P, _P, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P], _I),
    "nr_token_weights_fwd_pair": ([ctypes.POINTER(TokenWeightsProblem), ctypes.POINTER(TokenWeightsProblem), _I, _P], _I),
    "nr_token_weights_fwd_group": ([ctypes.POINTER(TokenWeightsProblem), ctypes.POINTER(ctypes.c_int), _I, _P], _I),
    "nr_token_weights_fwd": ([_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _P, _P, _P], _I),
    "nr_local_level_tiles": ([_I, _I, _I, _I, _I, ctypes.POINTER(_I), ctypes.POINTER(_I)], _I),
    "nr_local_level_fwd": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P], _I),
    "nr_local_level_group_kind": ([_I, _I, _I, _I, _I, _I], _I),
    "nr_local_level_group": ([_I, ctypes.POINTER(LocalLevelProblem), _P], _I),
    "nr_reduce_parts": ([_P, _I, _I, _F, _P, _P], _I),
    "nr_gemm_nt_f32": ([_P, _P, _I, _I, _I, _P, _P], _I),
    "nr_centrality_weights": ([_P, _I, _I, _P, _I, _I, _F, _P, _P, _P, _P], _I),
    "nr_centrality_weights_pair": ([_P, _P, _I, _I, _I, _I, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P], _I),
    "nr_dpc_workspace_bytes": ([_I, _I], _Z),
    "nr_dpc_knn_assign": ([_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P], _I),
    "nr_shift_concat": ([_P, _I, _I, _I, _P, _P], _I),
    "nr_ctm_norm_score": ([_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P], _I),
    "nr_merge_ln": ([_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _F, _P, _P, _P, _P], _I),
    "nr_tc_attention": ([_P, _P, _P, _I, _I, _I, _I, _I, _P, _P], _I),
    "nr_ctm_front": ([_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P], _I),
    "nr_linear_x3": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P], _I),
    "nr_shift_concat_split": ([_P, _I, _I, _I, _P, _P, _P], _I),
    "nr_ctm_back": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _F, _P, _P, _P, _P, _P], _I),
    "nr_ctm_stage_workspace_bytes": ([_I, _I, _I, _I], _Z),
    "nr_ctm_stage_workspace_layout": ([_I, _I, _I, _I, ctypes.POINTER(_Z)], _I),
    "nr_ctm_stage_workspace_layout2": ([_I, _I, _I, _I, ctypes.POINTER(_Z)], _I),
    "nr_split_group": ([_I, ctypes.POINTER(SplitItem), _P], _I),
    "nr_colsum_group": ([_I, ctypes.POINTER(ColsumItem), _P], _I),
    "nr_linear_group": ([_I, ctypes.POINTER(LinearProblem), _P], _I),
    "nr_ctm_attn_bwd": ([_I, ctypes.POINTER(CtmAttnBwdDesc), _P], _I),
    "nr_ctm_mid_bwd": ([_I, ctypes.POINTER(CtmMidBwdDesc), _P], _I),
    "nr_ctm_stage_fwd": ([ctypes.POINTER(CtmStageDesc), _I, _P], _I),
    "nr_ctm_stage_fwd_range": ([ctypes.POINTER(CtmStageDesc), _I, _I, _I, _P], _I),
    "nr_sinkhorn_workspace_bytes": ([_I], _Z),
    "nr_sinkhorn_cooperative_ok": ([_I], _I),
    "nr_sinkhorn_cooperative_gate": ([_I, _I, _I, _I], _I),
    "nr_sinkhorn_targets_multilaunch": ([_P, _I, _F, _I, _P, _P, _P, _P], _I),
    "nr_sinkhorn_uniform_rows": ([_P, _I, _F, _I, _F, _P, _I, _P, _P, _P, _P], _I),
    "nr_row_losses_fwd_no_uniform": ([_P] * 7 + [_I, _I, _F, _P, _P], _I),
    "nr_split_tail_workgroups": ([_I], _I),
    "nr_sinkhorn_uniform_rows_final": ([_P, _I, _F, _I, _F, _P, _P, _F, _F, _F, _P, _P, _P], _I),
    "nr_row_losses_fwd_no_uniform_final": ([_P, _P, _P, _I, _P, _I, _F, _P, _P, _P, _I, _I, _F, _P, _P, _F, _F, _F, _P, _P], _I),
    "nr_row_losses_fwd_no_uniform_final_cw": ([_P, _P, _P, _I, _P, _I, _F, _P, _P, _P, _P, _I, _F, _P, _I, _I, _F, _P, _P, _F, _F, _F, _P, _P], _I),
    "nr_sinkhorn_targets": ([_P, _I, _F, _I, _P, _P, _P, _P], _I),
    "nr_row_losses_fwd": ([_P] * 9 + [_I, _I, _F, _P, _P], _I),
    "nr_row_losses_fwd_slab": ([_P, _P, _I, _I] + [_P] * 8 + [_I, _I, _F, _P, _P], _I),
    "nr_row_losses_fwd_final": ([_P] * 9 + [_I, _I, _F, _P, _P, _F, _F, _F, _P, _P], _I),
    "nr_loss_finalize": ([_P, _I, _F, _F, _F, _P, _P], _I),
    "nr_row_losses_bwd": ([_P] * 9 + [_I, _I, _F, _P, _P, _P, _P, _P, _P, _P], _I),
    "nr_row_losses_bwd_slab": ([_P, _P, _I, _I] + [_P] * 8 + [_I, _I, _F, _P, _P, _P, _P, _P, _P, _P], _I),
    "nr_add_transposed": ([_P, _P, _I, _P, _P], _I),
    "nr_colsum": ([_P, _I, _I, _P, _P], _I),
    "nr_local_level_bwd_workspace_bytes": ([_I, _I, _I, _I, _I, _I], _Z),
    "nr_local_level_bwd": ([_I, _P, _I, _F] + [_P] * 8 + [_I, _I, _I, _I, _I, _P, _P, _I, _P, _P], _I),
    "nr_local_level_bwd_mfma_supported": ([_I, _I, _I], _I),
    "nr_local_level_bwd_mfma_workspace_bytes": ([_I, _I, _I, _I, _I, _I], _Z),
    "nr_local_level_bwd_mfma": ([_I, _P, _I, _F, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P, _P], _I),
    "nr_local_level_bwd_group_workspace_bytes": ([_I, ctypes.POINTER(SimBwdItem)], _Z),
    "nr_local_level_bwd_group": ([_I, ctypes.POINTER(SimBwdItem), _P, _Z, _P], _I),
    "nr_pool_weight_bwd_group": ([_I, ctypes.POINTER(PoolWJob), _P], _I),
    "nr_sim_bwd_operand_group": ([_I, ctypes.POINTER(SimBwdOperand), _P], _I),
    "nr_slab_sum_group": ([_I, ctypes.POINTER(SlabSum), _P], _I),
    "nr_rowloss_coef": ([_P, _P, _P, _P, _P, _F, _F, _F, _I, _P, _P], _I),
    "nr_rowloss_bwd_finish": ([_P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P], _I),
    "nr_centrality_weights_bwd_pair": ([_P] * 10 + [_I, _I, _F, _P, _P, _P, _P, _P], _I),
    "nr_global_logits_bwd": ([_P, _P, _P, _P, _P, _I, _I, _P, _P, _P], _I),
    "nr_normalize_bwd": ([_P, _P, _P, _P, _P, _I, _I, _P, _P], _I),
    "nr_token_softmax_bwd": ([_P, _P, _I, _I, _P, _P], _I),
    "nr_centrality_weights_bwd": ([_P, _P, _P, _P, _P, _I, _I, _F, _P, _P, _P], _I),
    "nr_pack_shard": ([_I, _P, _P, _P, _P, _P], _I),
    "nr_pack_shard_convert": ([_I, _P, _P, _P, _P, _P, _P], _I),
    "nr_copy_group": ([_I, _P, _P, _P, _P], _I),
    "nr_unpack_gathered": ([_I, _P, _I, _Z, _P, _P, _P, _P, _P], _I),
    "nr_allgather_packed": ([_P, _I, _I, _P, _P, _P, _Z, _P, _P, _P, _P, _P], _I),
    "nr_step_prologue": ([_P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _P], _I),
    "nr_bank_push": ([_P, _P, _I, _I, _Z, _P, _P], _I),
    "nr_bank_ring_push": ([_I, _P, _P, _P, _I, _I, _P, _I, _P], _I),
    "nr_bank_absorb_gathered": ([ctypes.POINTER(BankAbsorbDesc), _P], _I),
    "nr_bank_absorb_counter_words": ([], _I),
    "nr_diag_ranks": ([_P, _I, _P, _P, _P], _I),
    "nr_slab_ranks": ([_P, _I, _I, _I, _P, _P, _P, _P, _P, _P], _I),
    "nr_group_slab_ranks": ([_P, _I, _I, _I, _P, _I, _P, _P, _P, _P], _I),
}


class NrHipError(RuntimeError):
    pass


def lib():
    """The loaded library; raises if it has not been built (python -m neighborretr_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NrHipError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -m neighborretr_amd.build` or __graft_entry__.build()). "
                "There is no fallback path.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (argtypes, restype) in _SIGNATURES.items():
            if not hasattr(handle, name):
                continue          # declared in the header but not built yet -> call() raises
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = restype
        # the descriptor structs mirrored above must have the library's layout: a stale .so against a newer header (or the
        # other way round) is refused here instead of corrupting a grouped launch's arguments
        if hasattr(handle, "nr_struct_size"):
            for cname, cls in STRUCTS.items():
                want = int(handle.nr_struct_size(cname.encode()))
                if want != ctypes.sizeof(cls):
                    raise NrHipError(f"{LIB_PATH}: sizeof({cname}) is {want}, the Python mirror has {ctypes.sizeof(cls)} "
                                     "-- rebuild the extension (python -m neighborretr_amd.build --force)")
        if int(handle.nr_version()) != ABI_VERSION:
            raise NrHipError(f"{LIB_PATH}: C ABI version {int(handle.nr_version())}, this binding was written for {ABI_VERSION} -- rebuild "
                             "the extension (python -m neighborretr_amd.build --force)")
        _lib = handle
    return _lib


def exported_symbols():
    return list(_SIGNATURES)


def _check(name, rc):
    if rc != 0:
        kind = "invalid argument / unsupported shape" if rc < 0 else "hipError_t"
        raise NrHipError(f"{name} failed with status {rc} ({kind})")


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_RAW_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """The current stream's hipStream_t.  (torch.cuda.current_stream() builds a Stream object per call, ~10 us: at 40-80 C-ABI
    calls per step that was 0.4 ms of an eager training step's host time; the raw getter costs well under 1 us.)"""
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        return ctypes.c_void_p(_RAW_STREAM(_RAW_DEVICE()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=None, allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise NrHipError("required tensor argument is None")
    if not t.is_cuda:
        raise NrHipError("tensor is not on the GPU; the HIP path has no CPU fallback")
    if not t.is_contiguous():
        raise NrHipError("tensor must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise NrHipError(f"expected dtype {dtype}, got {t.dtype}")
    return ctypes.c_void_p(t.data_ptr())


N_CALLS = 0          # C-ABI entry-point calls through call() (bench.py reports the count of one step)


def call(name, *args):
    global N_CALLS
    N_CALLS += 1
    fn = getattr(lib(), name, None)
    if fn is None:
        raise NrHipError(f"{name} is not exported by {LIB_PATH}")
    _check(name, fn(*args))


def version():
    return lib().nr_version()


def stream_capture_id(stream=None):
    """Capture id of `stream` (default: the current stream); 0 when it is not capturing."""
    cid = ctypes.c_ulonglong(0)
    st = torch.cuda.current_stream() if stream is None else stream
    _check("nr_stream_capture_id", lib().nr_stream_capture_id(ctypes.c_void_p(st.cuda_stream), ctypes.byref(cid)))
    return int(cid.value)


def prepare_parts(n_tok):
    return lib().nr_prepare_parts(int(n_tok))


def local_level_tiles(A, Nt, Bv, Nv, prec=PREC_BF16):
    r, c = _I(0), _I(0)
    _check("nr_local_level_tiles", lib().nr_local_level_tiles(A, Nt, Bv, Nv, int(prec), ctypes.byref(r), ctypes.byref(c)))
    return r.value, c.value


def local_level_group_kind(A, Nt, Bv, Nv, d, prec):
    """>= 0 if nr_local_level_group takes this product, -1 if it has to be launched on its own."""
    return int(lib().nr_local_level_group_kind(int(A), int(Nt), int(Bv), int(Nv), int(d), int(prec)))


def sinkhorn_workspace_bytes(B):
    return int(lib().nr_sinkhorn_workspace_bytes(int(B)))
