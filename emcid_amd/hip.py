"""ctypes binding of the C-ABI kernel library (include/emcid_hip.h -> csrc/libemcid_hip.so).

There is NO fallback: if the library is missing or a call fails, the product path raises.
torch is used here only to hand over device pointers and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from pathlib import Path
from typing import Optional

import torch

_LIB_PATH = Path(__file__).resolve().parent / "csrc" / "libemcid_hip.so"
_lib = None

EXPORTS = [
    "emcid_abi_version", "emcid_last_error", "emcid_gram_accumulate_f32", "emcid_symmetrize_lower_f32",
    "emcid_gather_mean_f32", "emcid_edit_workspace_bytes", "emcid_edit_layer_f64", "emcid_assemble_spd_f64",
    "emcid_cholesky_f64", "emcid_cholesky_solve_f64", "emcid_delta_w_f64", "emcid_dgemm_f64", "emcid_dgemm_ex_f64", "emcid_dgemm_batched_f64", "emcid_streamk_workspace_bytes", "emcid_dgemm_streamk_f64", "emcid_debug_streamk_stamps", "emcid_debug_step_stamps", "emcid_debug_linear_sp16_stamps", "emcid_fingerprint_store", "emcid_fingerprint_check", "emcid_axpy_f32",
    "emcid_profile_enable", "emcid_profile_collect", "emcid_attention_f32", "emcid_edit_layer_shard_f64",
    "emcid_apply_update_f32", "emcid_inverse_workspace_doubles", "emcid_quick_gelu_f32", "emcid_add_layernorm_f32", "emcid_embed_layernorm_f32", "emcid_tree_attention_f32", "emcid_debug_leaf_stamps",
    "emcid_cov_factor_workspace_bytes", "emcid_factor_cov_f64", "emcid_cov_inverse_f64",
    "emcid_edit_dual_workspace_bytes",
    "emcid_edit_dual_stage1_f64", "emcid_edit_dual_pt", "emcid_edit_dual_stage2_f64",
    "emcid_edit_dual_apply_stage1_f64", "emcid_edit_dual_yt", "emcid_edit_dual_apply_stage2_f64",
    "emcid_edit_dual_apply_assemble_f64",
    "emcid_edit_lu_workspace_bytes", "emcid_edit_layer_lu_f64", "emcid_lu_solve_f64",
    "emcid_edit_dual_cols_stage1_f64", "emcid_edit_dual_s", "emcid_edit_dual_u", "emcid_edit_dual_cols_stage2_f64",
    "emcid_apply_update2d_f32", "emcid_linear_f32", "emcid_linear_ws_f32", "emcid_linear_workspace_bytes",
    "emcid_split_rows_f16", "emcid_linear_sp16_f32", "emcid_gram_sp16_workspace_bytes", "emcid_gram_sp16_workspace_bytes_for", "emcid_gram_accumulate_sp16_f32", "emcid_add_layernorm_sp16", "emcid_embed_layernorm_sp16",
    "emcid_tree_attention_sp16", "emcid_tree_attention_sp16_supported", "emcid_clip_workspace_bytes",
    "emcid_clip_layer_head_sp16", "emcid_clip_layer_tail_sp16", "emcid_clip_layers_sp16", "emcid_clip_edit_layer_tail_sp16",
]
PROF_CLASSES = ["prep", "assemble", "chol_leaf", "chol_panel", "chol_trail", "trsm_diag", "trsm_update", "delta_w",
                "gram", "gather", "dgemm", "misc", "inv_build", "chol_inner", "inv_apply", "inv_block", "linear"]

ABI_VERSION = 14
NB = 128      # Cholesky block (csrc/common.h)
NPAD = 64     # concept padding of the f64 stacks (csrc/common.h)


class EmcidHipError(RuntimeError):
    pass


def lib_path() -> Path:
    return _LIB_PATH


def load():
    """dlopen the kernel library (works without a GPU: used by the CPU symbol-export test)."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise EmcidHipError(
            f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C {_LIB_PATH.parent}`. The MI355X path has no CPU fallback.")
    lib = C.CDLL(str(_LIB_PATH))
    p, i64, i32, f64, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_float
    sig = {
        "emcid_abi_version": (i32, []),
        "emcid_last_error": (C.c_char_p, []),
        "emcid_gram_accumulate_f32": (i32, [p, i64, i64, i64, p, i64, i32, p]),
        "emcid_symmetrize_lower_f32": (i32, [p, i64, i64, p]),
        "emcid_gather_mean_f32": (i32, [p, i64, i64, i64, i64, i64, p, p, i64, p, i64, p]),
        "emcid_edit_workspace_bytes": (i64, [i64, i64, i64]),
        "emcid_edit_layer_f64": (i32, [p, p, p, p, i64, i64, i64, f64, f64, i32, p, p, p, p, p, p, i64, p, p]),
        "emcid_edit_layer_shard_f64": (i32, [p, p, p, p, i64, i64, i64, f64, f64, i32, i64, i64, p, p, p, p, i64, p, p]),
        "emcid_apply_update_f32": (i32, [p, p, p, p, i64, p]),
        "emcid_assemble_spd_f64": (i32, [p, i64, p, i64, i64, i64, f64, f32, p, i64, p]),
        "emcid_cholesky_f64": (i32, [p, p, i64, i64, p, p, p]),
        "emcid_inverse_workspace_doubles": (i64, [i64]),
        "emcid_debug_leaf_stamps": (i32, [p, p, p, p, p, p]),
        "emcid_cov_factor_workspace_bytes": (i64, [i64, i64]),
        "emcid_factor_cov_f64": (i32, [p, i64, i64, f64, f64, p, i64, p, p]),
        "emcid_cov_inverse_f64": (i32, [p, i64, i64, i64, i64, p]),
        "emcid_edit_dual_workspace_bytes": (i64, [i64, i64, i64]),
        "emcid_edit_dual_stage1_f64": (i32, [p, p, p, i64, i64, i64, f64, i32, f64, p, i64, i64, i64, i64, i32, p, i64, p]),
        "emcid_edit_dual_pt": (p, [p, i64, i64, i64]),
        "emcid_edit_dual_stage2_f64": (i32, [i64, i64, i64, f64, p, p, p, p, p, p, i64, p, p]),
        "emcid_edit_dual_apply_stage1_f64": (i32, [p, p, p, i64, i64, i64, f64, i32, f64, p, i64, i64, i64, i64, i32, p, i64, p]),
        "emcid_edit_dual_yt": (p, [p, i64, i64, i64]),
        "emcid_edit_dual_apply_stage2_f64": (i32, [i64, i64, i64, p, i64, i64, i32, i32, p, p, p, p, i64, p, p]),
        "emcid_edit_dual_apply_assemble_f64": (i32, [i64, i64, i64, p, i64, p]),
        "emcid_cholesky_solve_f64": (i32, [p, i64, i64, p, p, p, i64, i64, p]),
        "emcid_edit_dual_cols_stage1_f64": (i32, [p, p, p, i64, i64, i64, f64, i32, f64, p, i64, i64, p, i32, p, i64, p]),
        "emcid_edit_dual_s": (p, [p, i64, i64, i64]),
        "emcid_edit_dual_u": (p, [p, i64, i64, i64]),
        "emcid_edit_dual_cols_stage2_f64": (i32, [i64, i64, i64, p, i64, i64, p, i32, p, i64, p, p]),
        "emcid_apply_update2d_f32": (i32, [p, i64, p, p, p, i64, i64, p]),
        "emcid_edit_lu_workspace_bytes": (i64, [i64, i64, i64]),
        "emcid_edit_layer_lu_f64": (i32, [p, p, p, p, i64, i64, i64, f64, f64, i32, p, p, p, p, p, p, i64, p, p]),
        "emcid_lu_solve_f64": (i32, [p, i64, i64, p, i64, i64, p, p, p]),
        "emcid_delta_w_f64": (i32, [p, i64, p, i64, i64, i64, i64, p, p, i64, p, p, p]),
        "emcid_dgemm_f64": (i32, [i32, i32, i64, i64, i64, f64, p, i64, p, i64, f64, p, i64, p]),
        "emcid_dgemm_ex_f64": (i32, [i32, i32, i64, i64, i64, f64, p, i64, p, i64, f64, p, i64, i32, i32, i32, p]),
        "emcid_dgemm_batched_f64": (i32, [i32, i32, i64, i64, i64, f64, p, i64, i64, p, i64, i64, f64, p, i64, i64, i64, p]),
        "emcid_streamk_workspace_bytes": (i64, [i32]),
        "emcid_debug_streamk_stamps": (i32, [p]),
        "emcid_debug_linear_sp16_stamps": (i32, [p]),
        "emcid_fingerprint_store": (i32, [i64, p, p, p, p, i64, p]),
        "emcid_fingerprint_check": (i32, [p, i64, i64, i64, p, p, p]),
        "emcid_debug_step_stamps": (i32, [p]),
        "emcid_dgemm_streamk_f64": (i32, [i32, i64, i64, i64, f64, p, i64, p, i64, p, i64, i32, i32, f64, p, i64, p]),
        "emcid_axpy_f32": (i32, [p, p, i64, p]),
        "emcid_quick_gelu_f32": (i32, [p, p, i64, p]),
        "emcid_linear_f32": (i32, [p, i64, p, i64, p, p, i64, p, i64, i64, i64, i64, i32, i32, p]),
        "emcid_linear_ws_f32": (i32, [p, i64, p, i64, p, p, i64, p, i64, i64, i64, i64, i32, i32, p, i64, p]),
        "emcid_linear_workspace_bytes": (i64, []),
        "emcid_split_rows_f16": (i32, [p, i64, i64, i64, p, i64, p, p, p]),
        "emcid_gram_sp16_workspace_bytes": (i64, [i64]),
        "emcid_gram_sp16_workspace_bytes_for": (i64, [i64, i64]),
        "emcid_gram_accumulate_sp16_f32": (i32, [p, p, i64, i64, i64, p, i64, p, i64, p]),
        "emcid_add_layernorm_sp16": (i32, [p, i64, p, i64, p, p, C.c_float, i64, i64, p, p, p, i64, p, p, p, p]),
        "emcid_embed_layernorm_sp16": (i32, [p, i64, i64, p, i64, i64, p, p, p, p, C.c_float, i64, i64, p, p, p, i64, p, p]),
        "emcid_tree_attention_sp16_supported": (i32, [i64, i64, i64]),
        "emcid_clip_workspace_bytes": (i64, [i64, i64, i64]),
        "emcid_clip_layer_head_sp16": (i32, [p, i64, i64, i64, i64, f32, p, i64, p, p, i64, p, p, p, p, p, p, p, p, i64, p]),
        "emcid_clip_layer_tail_sp16": (i32, [p, i64, i64, i64, p, p, p, p, p, p, f32, p, p, p]),
        "emcid_clip_layers_sp16": (i32, [p, i64, i64, i64, i64, i64, f32, p, i64, p, p, p, p, p, p, f32, p, i64, p]),
        "emcid_clip_edit_layer_tail_sp16": (i32, [p, i64, i64, i64, p, p, p, p, p, p, i64, i64, p, f64, i32, f64, p, i64, i64, i32,
                                                 p, p, p, p, p, p, p, p, i64, p, p, i64, p, p, p, f32, p, p, p]),
        "emcid_tree_attention_sp16": (i32, [p, i64, p, p, i64, p, i64, p, p, i64, i64, i64, f32, p, i64, p, p]),
        "emcid_linear_sp16_f32": (i32, [p, i64, p, p, i64, p, p, p, i64, p, i64, p, i64, p, i64, i64, i64, i32, i32, p]),
        "emcid_add_layernorm_f32": (i32, [p, i64, p, i64, p, p, C.c_float, i64, i64, p, p, p]),
        "emcid_embed_layernorm_f32": (i32, [p, i64, i64, p, i64, i64, p, p, p, p, C.c_float, i64, i64, p, p, p]),
        "emcid_tree_attention_f32": (i32, [p, i64, p, p, i64, p, i64, p, p, i64, i64, i64, f32, p, i64, p]),
        "emcid_profile_enable": (i32, [C.c_uint]),
        "emcid_profile_collect": (i32, [p, p, i32]),
        "emcid_attention_f32": (i32, [p, p, p, i64, i64, i64, p, i32, i64, i64, i32, f32, i64, i64, i64, i64, p, p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.emcid_abi_version() != ABI_VERSION:
        raise EmcidHipError(f"ABI mismatch: library {lib.emcid_abi_version()} vs binding {ABI_VERSION}; rebuild csrc/")
    _lib = lib
    return lib


_tls = threading.local()  # .restore: device ordinals to go back to after the call in flight (see _stream), per thread


def _check(rc: int, what: str):
    restore = getattr(_tls, "restore", None)
    if restore:
        torch.cuda.set_device(restore.pop())
    if rc != 0:
        raise EmcidHipError(f"{what} failed (rc={rc}): {load().emcid_last_error().decode()}")


def _ptr(t: Optional[torch.Tensor], dtype=None, what="tensor"):
    if t is None:
        return None
    if not t.is_cuda:
        raise EmcidHipError(f"{what} must live in HBM (got device {t.device}); there is no CPU path")
    if dtype is not None and t.dtype != dtype:
        raise EmcidHipError(f"{what} must be {dtype}, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


def _stream(t: torch.Tensor):
    """The current HIP stream of the tensor's device — and that device made current for the duration of the call
    (include/emcid_hip.h: the library keys its capture streams, cached graphs and launches by the current device; a model
    on cuda:1 while cuda:0 is current must still run on cuda:1).  Every call site is ``_check(lib.fn(..., _stream(t)), name)``:
    ``_check`` switches back."""
    idx = t.device.index
    cur = torch.cuda.current_device()
    if idx is not None and idx != cur:
        if getattr(_tls, "restore", None) is None:
            _tls.restore = []
        _tls.restore.append(cur)
        torch.cuda.set_device(idx)
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


# ---- Stage 0 -----------------------------------------------------------------------------------------

GRAM_SPLIT = os.environ.get("EMCID_GRAM_SPLIT", "1") != "0"      # 0: always the exact-f32 MFMA SYRK
_GRAM_WS = {}


def gram_takes_row_weight(X: torch.Tensor, ksplit: int = 0) -> bool:
    """whether gram_accumulate_ would run this batch on the split-fp16 path (which applies a per-row weight as it reads the rows)"""
    d = X.shape[1]
    return bool(GRAM_SPLIT and ksplit != 1 and X.shape[0] >= 2048 and d % 4 == 0 and d >= 256 and X.stride(0) % 4 == 0
                and X.stride(1) == 1 and X.data_ptr() % 16 == 0 and X.dtype == torch.float32)


def gram_accumulate_(G: torch.Tensor, X: torch.Tensor, ksplit: int = 0, row_weight: Optional[torch.Tensor] = None):
    """G (d,d) fp32 lower triangle += X^T X, X (t,d) fp32 with row stride multiple of 4 (runningstats.py:493).  Long batches
    (>= 2048 tokens, ``ksplit != 1``) run on the split-fp16 path (csrc/gemm_sp16.hip: three f16 MFMAs per k-step on X^T planes
    under per-feature scales, fp32 atomics into G); ``ksplit == 1`` — the deterministic mode — and short batches on the exact-f32
    SYRK (csrc/gram_f32.hip).  ``row_weight`` (t,) fp32: row r enters as fl32(row_weight[r] * X[r]) — on the split path inside the
    kernels that read the rows, otherwise by a multiplication here."""
    assert X.dim() == 2 and G.dim() == 2 and G.shape[0] == G.shape[1] == X.shape[1]
    assert X.stride(1) == 1 and G.stride(1) == 1
    if X.shape[0] == 0:
        return G
    d = X.shape[1]
    if row_weight is not None:
        assert row_weight.shape == (X.shape[0],) and row_weight.dtype == torch.float32 and row_weight.is_contiguous()
        if not (gram_takes_row_weight(X, ksplit) and G.dtype == torch.float32):
            X, row_weight = X * row_weight.unsqueeze(1), None
    if GRAM_SPLIT and ksplit != 1 and X.shape[0] >= 2048 and d % 4 == 0 and d >= 256 and X.stride(0) % 4 == 0 \
            and X.data_ptr() % 16 == 0 and X.dtype == torch.float32 and G.dtype == torch.float32:
        key = (X.device.index if X.device.index is not None else torch.cuda.current_device(),
               torch.cuda.current_stream(X.device).cuda_stream, d)
        # one workspace per (device, stream, d), sized for the largest batch seen there (403 MB at d = 3072 only when a batch
        # reaches the 32 768-token chunk; a 2 048-token batch takes 25 MB)
        need = int(load().emcid_gram_sp16_workspace_bytes_for(d, X.shape[0]))
        ws = _GRAM_WS.get(key)
        if ws is None or ws.numel() < need:
            if ws is None and len(_GRAM_WS) >= 8:
                _GRAM_WS.clear()
            ws = _GRAM_WS[key] = torch.empty(need, dtype=torch.uint8, device=X.device)
        _check(load().emcid_gram_accumulate_sp16_f32(_ptr(X, torch.float32, "X"),
                                                     _ptr(row_weight, torch.float32, "row_weight") if row_weight is not None else None,
                                                     X.shape[0], d, X.stride(0),
                                                     _ptr(G, torch.float32, "G"), G.stride(0), C.c_void_p(ws.data_ptr()),
                                                     ws.numel(), _stream(G)), "emcid_gram_accumulate_sp16_f32")
        return G
    _check(load().emcid_gram_accumulate_f32(_ptr(X, torch.float32, "X"), X.shape[0], X.shape[1], X.stride(0),
                                            _ptr(G, torch.float32, "G"), G.stride(0), ksplit, _stream(G)),
           "emcid_gram_accumulate_f32")
    return G


def symmetrize_lower_(G: torch.Tensor):
    _check(load().emcid_symmetrize_lower_f32(_ptr(G, torch.float32, "G"), G.shape[0], G.stride(0), _stream(G)),
           "emcid_symmetrize_lower_f32")
    return G


# ---- K/Z assembly -------------------------------------------------------------------------------------

def gather_mean(act: torch.Tensor, idx: torch.Tensor, seg: torch.Tensor, out: Optional[torch.Tensor] = None):
    """act (B,S,c) fp32, idx (B,) int64 token position per prompt, seg (N+1,) int64 prompt offsets per request."""
    assert act.dim() == 3 and act.stride(2) == 1
    B, S, c = act.shape
    N = seg.numel() - 1
    assert idx.numel() == B
    if out is None:
        out = torch.empty(N, c, dtype=torch.float32, device=act.device)
    _check(load().emcid_gather_mean_f32(_ptr(act, torch.float32, "act"), B, S, c, act.stride(0), act.stride(1),
                                        _ptr(idx, torch.int64, "idx"), _ptr(seg, torch.int64, "seg"), N,
                                        _ptr(out, torch.float32, "out"), out.stride(0), _stream(act)),
           "emcid_gather_mean_f32")
    return out


# ---- Stage 2 ------------------------------------------------------------------------------------------

def edit_workspace_bytes(N: int, d: int, h: int) -> int:
    return int(load().emcid_edit_workspace_bytes(N, d, h))


class EditWorkspace:
    """Reusable HBM workspace (+ the device `info` word) for emcid_edit_layer_f64."""

    def __init__(self, N: int, d: int, h: int, device):
        self.key = (N, d, h)
        self.nbytes = edit_workspace_bytes(N, d, h)
        self.buf = torch.zeros(self.nbytes // 8, dtype=torch.float64, device=device)      # (zero: padding the stages never write)
        self.info = torch.zeros(1, dtype=torch.int32, device=device)


def edit_layer(K, Zc, zs_t, Cov, lam: float, edit_weight: float, layers_left: int, W0=None, W=None,
               want_factors: bool = False, want_dw: bool = True, ws: Optional[EditWorkspace] = None):
    """One edited layer of emcid_main.py:1016-1061 on the GPU.  Returns dict(Xt, Rt, dW) (entries may be None)."""
    N, d = K.shape
    h = Zc.shape[1]
    assert Cov.shape == (d, d) and zs_t.shape == (N, h) and Zc.shape == (N, h)
    for t, nm in ((K, "K"), (Zc, "Zc"), (zs_t, "zs_t"), (Cov, "C")):
        assert t.is_contiguous(), nm
    if ws is None or ws.key != (N, d, h):
        ws = EditWorkspace(N, d, h, K.device)
    dev = K.device
    Xt = torch.empty(N, d, dtype=torch.float64, device=dev) if want_factors else None
    Rt = torch.empty(N, h, dtype=torch.float64, device=dev) if want_factors else None
    dW = torch.empty(h, d, dtype=torch.float32, device=dev) if want_dw else None
    if W is not None:
        assert W.is_contiguous() and W.shape == (h, d) and W0 is not None and W0.is_contiguous()
    _check(load().emcid_edit_layer_f64(
        _ptr(K, torch.float32, "K"), _ptr(Zc, torch.float32, "Zc"), _ptr(zs_t, torch.float32, "zs_t"),
        _ptr(Cov, torch.float32, "C"), N, d, h, float(lam), float(edit_weight), int(layers_left),
        _ptr(W0, torch.float32, "W0"), _ptr(W, torch.float32, "W"), _ptr(Xt), _ptr(Rt), _ptr(dW),
        _ptr(ws.buf), ws.nbytes, _ptr(ws.info, torch.int32, "info"), _stream(K)), "emcid_edit_layer_f64")
    return {"Xt": Xt, "Rt": Rt, "dW": dW, "ws": ws}


class LuWorkspace:
    """Workspace of the pivoted-LU fallback (emcid_edit_layer_lu_f64)."""

    def __init__(self, N: int, d: int, h: int, device):
        self.key = (N, d, h)
        self.nbytes = int(load().emcid_edit_lu_workspace_bytes(N, d, h))
        self.buf = torch.zeros(self.nbytes // 8, dtype=torch.float64, device=device)      # (zero: padding the stages never write)
        self.info = torch.zeros(1, dtype=torch.int32, device=device)


def edit_layer_lu(K, Zc, zs_t, Cov, lam: float, edit_weight: float, layers_left: int, W0=None, W=None,
                  want_factors: bool = False, want_dw: bool = True, ws: Optional[LuWorkspace] = None):
    """One edited layer solved like the reference does (LU with partial pivoting, torch.linalg.solve's algorithm): works for
    any nonsingular lam*C' + K K^T.  Returns dict(adj_k (d, N) | None, Rt (N, h) | None, dW, ws)."""
    N, d = K.shape
    h = Zc.shape[1]
    assert Cov.shape == (d, d) and zs_t.shape == (N, h) and Zc.shape == (N, h)
    for t, nm in ((K, "K"), (Zc, "Zc"), (zs_t, "zs_t"), (Cov, "C")):
        assert t.is_contiguous(), nm
    if ws is None or ws.key != (N, d, h):
        ws = LuWorkspace(N, d, h, K.device)
    dev = K.device
    adj_k = torch.empty(d, N, dtype=torch.float64, device=dev) if want_factors else None
    Rt = torch.empty(N, h, dtype=torch.float64, device=dev) if want_factors else None
    dW = torch.empty(h, d, dtype=torch.float32, device=dev) if want_dw else None
    if W is not None:
        assert W.is_contiguous() and W.shape == (h, d) and W0 is not None and W0.is_contiguous()
    _check(load().emcid_edit_layer_lu_f64(
        _ptr(K, torch.float32, "K"), _ptr(Zc, torch.float32, "Zc"), _ptr(zs_t, torch.float32, "zs_t"),
        _ptr(Cov, torch.float32, "C"), N, d, h, float(lam), float(edit_weight), int(layers_left),
        _ptr(W0, torch.float32, "W0"), _ptr(W, torch.float32, "W"), _ptr(adj_k), _ptr(Rt), _ptr(dW),
        _ptr(ws.buf), ws.nbytes, _ptr(ws.info, torch.int32, "info"), _stream(K)), "emcid_edit_layer_lu_f64")
    return {"adj_k": adj_k, "Rt": Rt, "dW": dW, "ws": ws}


def lu_solve_(A: torch.Tensor, B: torch.Tensor):
    """Test hook: solves A X = B in place (A (n, n) f64 becomes P L U, B (n, nrhs) f64 becomes X) by LU with partial
    pivoting; returns (pivots int32 (n,), info int32 (1,))."""
    n = A.shape[0]
    piv = torch.zeros(n, dtype=torch.int32, device=A.device)
    info = torch.zeros(1, dtype=torch.int32, device=A.device)
    _check(load().emcid_lu_solve_f64(_ptr(A, torch.float64, "A"), A.stride(0), n, _ptr(B, torch.float64, "B"), B.stride(0),
                                     B.shape[1], _ptr(piv), _ptr(info), _stream(A)), "emcid_lu_solve_f64")
    return piv, info


def edit_layer_shard(K, Zc, zs_t, Cov, lam: float, edit_weight: float, layers_left: int, rows, want_factors=False,
                     ws: Optional[EditWorkspace] = None):
    """Concept-sharded layer: all N concepts assemble/factor A, only rows [lo, hi) are solved.
    Returns dict(U=(h,d) f64 partial, Xt, Rt (shard rows or None), ws)."""
    N, d = K.shape
    h = Zc.shape[1]
    lo, hi = rows
    assert Cov.shape == (d, d) and zs_t.shape == (N, h) and Zc.shape == (N, h) and 0 <= lo < hi <= N
    for t, nm in ((K, "K"), (Zc, "Zc"), (zs_t, "zs_t"), (Cov, "C")):
        assert t.is_contiguous(), nm
    if ws is None or ws.key != (N, d, h):
        ws = EditWorkspace(N, d, h, K.device)
    dev = K.device
    U = torch.empty(h, d, dtype=torch.float64, device=dev)
    Xt = torch.empty(hi - lo, d, dtype=torch.float64, device=dev) if want_factors else None
    Rt = torch.empty(hi - lo, h, dtype=torch.float64, device=dev) if want_factors else None
    _check(load().emcid_edit_layer_shard_f64(
        _ptr(K, torch.float32, "K"), _ptr(Zc, torch.float32, "Zc"), _ptr(zs_t, torch.float32, "zs_t"),
        _ptr(Cov, torch.float32, "C"), N, d, h, float(lam), float(edit_weight), int(layers_left), lo, hi,
        _ptr(U), _ptr(Xt), _ptr(Rt), _ptr(ws.buf), ws.nbytes, _ptr(ws.info, torch.int32, "info"), _stream(K)),
        "emcid_edit_layer_shard_f64")
    return {"U": U, "Xt": Xt, "Rt": Rt, "ws": ws}


def apply_update_(U, W0, W, want_dw=True):
    """W = W0 + float(U); returns dW = float(U) (or None)."""
    assert U.is_contiguous() and W.is_contiguous() and W0.is_contiguous() and U.shape == W.shape
    dW = torch.empty_like(W) if want_dw else None
    _check(load().emcid_apply_update_f32(_ptr(U, torch.float64, "U"), _ptr(W0, torch.float32, "W0"),
                                         _ptr(W, torch.float32, "W"), _ptr(dW), W.numel(), _stream(W)),
           "emcid_apply_update_f32")
    return dW


def dgemm(ta: int, tb: int, A, B, Cm, alpha=1.0, beta=0.0, M=None, N=None, K=None):
    """Test hook for the fp64 MFMA GEMM. ta/tb = 0: operand stored [rows][K]; 1: stored [K][rows]."""
    M = M if M is not None else (A.shape[0] if ta == 0 else A.shape[1])
    K = K if K is not None else (A.shape[1] if ta == 0 else A.shape[0])
    N = N if N is not None else (B.shape[0] if tb == 0 else B.shape[1])
    _check(load().emcid_dgemm_f64(ta, tb, M, N, K, float(alpha), _ptr(A, torch.float64), A.stride(0),
                                  _ptr(B, torch.float64), B.stride(0), float(beta), _ptr(Cm, torch.float64),
                                  Cm.stride(0), _stream(Cm)), "emcid_dgemm_f64")
    return Cm


def dgemm_ex(ta: int, tb: int, A, B, Cm, alpha=1.0, beta=0.0, flags=0, cfg=-1, ksplit=0):
    """Test hook: the GEMM with the solver's structure hints (triangular operands, lower-only, tile config, K split)."""
    M = A.shape[0] if ta == 0 else A.shape[1]
    K = A.shape[1] if ta == 0 else A.shape[0]
    N = B.shape[0] if tb == 0 else B.shape[1]
    _check(load().emcid_dgemm_ex_f64(ta, tb, M, N, K, float(alpha), _ptr(A, torch.float64), A.stride(0),
                                     _ptr(B, torch.float64), B.stride(0), float(beta), _ptr(Cm, torch.float64),
                                     Cm.stride(0), int(flags), int(cfg), int(ksplit), _stream(Cm)), "emcid_dgemm_ex_f64")
    return Cm


_streamk_ws = {}


def dgemm_streamk(tb: int, A, B, Cm, alpha=1.0, flags=1, wgs=256, diag_add=0.0):
    """Test hook: the two-phase (atomic-free, reproducible) stream-K GEMM; flags 1/2: triangular B, 16: lower-only square."""
    M, K = A.shape
    N = B.shape[0] if tb == 0 else B.shape[1]
    key = (str(A.device), wgs)
    ws = _streamk_ws.get(key)
    if ws is None:
        ws = _streamk_ws[key] = torch.zeros(int(load().emcid_streamk_workspace_bytes(wgs)) // 8, dtype=torch.float64, device=A.device)
    _check(load().emcid_dgemm_streamk_f64(tb, M, N, K, float(alpha), _ptr(A, torch.float64), A.stride(0), _ptr(B, torch.float64),
                                          B.stride(0), _ptr(Cm, torch.float64), Cm.stride(0), int(flags), int(wgs), float(diag_add),
                                          _ptr(ws), ws.numel() * 8, _stream(Cm)), "emcid_dgemm_streamk_f64")
    return Cm


def dgemm_batched(ta: int, tb: int, A, B, Cm, alpha=1.0, beta=0.0):
    """C[b] = alpha * opA(A[b]) opB(B[b]) + beta * C[b] for 3-D operands (batch, rows, cols) with unit inner stride;
    ta/tb as in `dgemm`.  A batch stride of 0 (an expanded operand) shares that operand across the batch."""
    nb = Cm.shape[0]
    M = A.shape[1] if ta == 0 else A.shape[2]
    K = A.shape[2] if ta == 0 else A.shape[1]
    N = B.shape[1] if tb == 0 else B.shape[2]
    for t_, nm in ((A, "A"), (B, "B"), (Cm, "C")):
        if t_.dim() != 3 or t_.stride(2) != 1 or t_.shape[0] != nb:
            raise EmcidHipError(f"dgemm_batched: {nm} must be (batch, rows, cols) with unit inner stride")
    if Cm.shape[1] != M or Cm.shape[2] != N or (B.shape[2] if tb == 0 else B.shape[1]) != K:
        raise EmcidHipError("dgemm_batched: shapes do not agree")
    _check(load().emcid_dgemm_batched_f64(ta, tb, M, N, K, float(alpha), _ptr(A, torch.float64), A.stride(1), A.stride(0),
                                          _ptr(B, torch.float64), B.stride(1), B.stride(0), float(beta),
                                          _ptr(Cm, torch.float64), Cm.stride(1), Cm.stride(0), nb, _stream(Cm)),
           "emcid_dgemm_batched_f64")
    return Cm


def fingerprint_store(entries, table: torch.Tensor):
    """Stale-cache guard (include/emcid_hip.h): leave {address, bytes, fingerprint of the bytes} of every ``(tensor, slot)`` of
    ``entries`` in ``table[slot]`` — one launch per 32 entries."""
    n = len(entries)
    if not n:
        return
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t, _ in entries])
    sizes = (C.c_int64 * n)(*[t.numel() * t.element_size() for t, _ in entries])
    slots = (C.c_int64 * n)(*[int(s) for _, s in entries])
    _check(load().emcid_fingerprint_store(n, ptrs, sizes, slots, _ptr(table, torch.int64, "table"), table.shape[0], _stream(table)),
           "emcid_fingerprint_store")


def fingerprint_check(table: torch.Tensor, first_slot: int, n_slots: int, flag: torch.Tensor, skip=()):
    """Recompute the fingerprints of ``table[first_slot : first_slot + n_slots]`` (empty slots and the slot offsets in ``skip``
    are not looked at); ``flag |= 1`` on a mismatch."""
    mask = (C.c_uint64 * 4)()
    for k in skip:
        mask[k >> 6] |= 1 << (k & 63)
    _check(load().emcid_fingerprint_check(_ptr(table, torch.int64, "table"), table.shape[0], int(first_slot), int(n_slots), mask,
                                          _ptr(flag, torch.int32, "flag"), _stream(table)), "emcid_fingerprint_check")


def axpy_(W: torch.Tensor, dW: torch.Tensor):
    """W += dW, fp32 (emcid_main.py:809)."""
    assert W.is_contiguous() and dW.is_contiguous() and W.numel() == dW.numel()
    _check(load().emcid_axpy_f32(_ptr(W, torch.float32, "W"), _ptr(dW, torch.float32, "dW"), W.numel(), _stream(W)),
           "emcid_axpy_f32")
    return W


def cholesky(A: torch.Tensor):
    """Test hook: A (dp,dp) f64 contiguous, dp % 128 == 0.  Returns (L, inverse workspace, info); the first
    ceil(dp/512) slots of 512*512 doubles of the workspace are the inverted 512-blocks of L (ld 512)."""
    dp = A.shape[0]
    L = torch.zeros_like(A)
    inv = torch.empty(int(load().emcid_inverse_workspace_doubles(dp)), dtype=torch.float64, device=A.device)
    info = torch.zeros(1, dtype=torch.int32, device=A.device)
    _check(load().emcid_cholesky_f64(_ptr(A, torch.float64), _ptr(L), dp, A.stride(0), _ptr(inv), _ptr(info), _stream(A)),
           "emcid_cholesky_f64")
    return L, inv, info


def cholesky_solve_(L, inv, Bt):
    """Test hook: Bt (Np,dp) f64 := Bt (L L^T)^{-1}."""
    Y = torch.empty_like(Bt)
    _check(load().emcid_cholesky_solve_f64(_ptr(L, torch.float64), L.shape[0], L.stride(0), _ptr(inv), _ptr(Bt), _ptr(Y),
                                           Bt.shape[0], Bt.stride(0), _stream(Bt)), "emcid_cholesky_solve_f64")
    return Bt


def delta_w_(Rt, Xt, W0, W):
    """W = W0 + float(Rt^T Xt): the dW contraction of emcid_main.py:1050,:1061 alone, for factors that are already
    there (Rt (N, h) f64 residual rows, Xt (N, d) f64 rows of adj_k^T; both contiguous)."""
    N, h = Rt.shape
    d = Xt.shape[1]
    assert Xt.shape[0] == N and W.shape == (h, d) and W0.shape == (h, d)
    for t, nm in ((Rt, "Rt"), (Xt, "Xt"), (W0, "W0"), (W, "W")):
        assert t.is_contiguous(), nm
    if h % 2 or d % 2:
        raise EmcidHipError("delta_w_: h and d must be even (16-byte aligned f64 rows)")
    _check(load().emcid_delta_w_f64(_ptr(Rt, torch.float64, "Rt"), h, _ptr(Xt, torch.float64, "Xt"), d, N, h, d,
                                    _ptr(W0, torch.float32, "W0"), _ptr(W, torch.float32, "W"), d, None, None, _stream(W)),
           "emcid_delta_w_f64")
    return W


def profile_enable(classes=()):
    """Start (or, with no classes, stop) HIP-event timing of the named kernel classes (PROF_CLASSES)."""
    mask = 0
    for c in classes:
        mask |= 1 << PROF_CLASSES.index(c)
    _check(load().emcid_profile_enable(mask), "emcid_profile_enable")


def profile_collect():
    """{class: (total_ms, launches)} for the classes that recorded launches since profile_enable."""
    n = len(PROF_CLASSES)
    ms = (C.c_double * n)()
    cnt = (C.c_int64 * n)()
    _check(load().emcid_profile_collect(ms, cnt, n), "emcid_profile_collect")
    return {PROF_CLASSES[i]: (ms[i], int(cnt[i])) for i in range(n) if cnt[i]}


def attention(q, k, v, mask=None, causal=False, scale=None):
    """q/k/v: (B, H, S, D) fp32 views with unit stride over D and identical strides; mask: None, bool keep-mask or
    additive float, shape (B|1, 1, S, S).  Returns (B, S, H, D) contiguous."""
    B, H, S, D = q.shape
    for t in (q, k, v):
        if t.stride(3) != 1 or t.stride() != q.stride() or t.shape != q.shape:
            raise EmcidHipError("attention: q/k/v must share shape/strides with unit stride over head_dim")
    out = torch.empty(B, S, H, D, dtype=torch.float32, device=q.device)
    kind, mptr, mb, mi = 0, None, 0, 0
    if mask is not None:
        if mask.dim() != 4 or mask.shape[1] != 1 or mask.shape[2] != S or mask.shape[3] != S or mask.stride(3) != 1:
            mask = mask.expand(-1, 1, S, S).contiguous() if mask.dim() == 4 and mask.shape[1] == 1 else None
            if mask is None:
                raise EmcidHipError("attention: mask must be (B|1, 1, S, S)")
        if mask.dtype == torch.bool:
            kind, mptr = 1, _ptr(mask)
        elif mask.dtype == torch.float32:
            kind, mptr = 2, _ptr(mask)
        else:
            raise EmcidHipError(f"attention: mask dtype {mask.dtype} not supported")
        mb = mask.stride(0) if mask.shape[0] > 1 else 0
        mi = mask.stride(2)
    scale = float(D ** -0.5 if scale is None else scale)
    _check(load().emcid_attention_f32(_ptr(q, torch.float32, "q"), _ptr(k, torch.float32, "k"), _ptr(v, torch.float32, "v"),
                                      q.stride(0), q.stride(1), q.stride(2), mptr, kind, mb, mi, int(bool(causal)), scale,
                                      B, H, S, D, _ptr(out), _stream(q)), "emcid_attention_f32")
    return out


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    """x * sigmoid(1.702 x), fp32, one pass."""
    x = x.contiguous()
    y = torch.empty_like(x)
    _check(load().emcid_quick_gelu_f32(_ptr(x, torch.float32, "x"), _ptr(y), x.numel(), _stream(x)), "emcid_quick_gelu_f32")
    return y


ACT_NONE, ACT_QUICK_GELU, ACT_GELU_ERF = 0, 1, 2
LINEAR_FLOPS = {"count": False, "flops": 0.0, "launches": 0}      # bench.py's roofline pass: algorithmic 2 M N K of every launch


def linear_supported(x: torch.Tensor, w: torch.Tensor) -> bool:
    """True when ``linear`` takes these operands: fp32 in HBM, K contiguous and a multiple of 16, 16-byte aligned rows."""
    return (x.is_cuda and w.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.dim() == 2 and w.dim() == 2
            and x.shape[1] == w.shape[1] and x.shape[1] % 16 == 0 and x.stride(1) == 1 and w.stride(1) == 1
            and x.stride(0) % 4 == 0 and w.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0
            and x.shape[0] > 0)


_LINEAR_WS = {}       # (device index, stream handle) -> the split-K form's workspace (zeroed once; every launch leaves its counters zero)


def linear_split_cfg(parts: int, prefetch: int = 2) -> int:
    """``cfg`` of the split-K form: 128 x 128 tiles, four waves, every tile's K range over ``parts`` (2..8) workgroups."""
    return 64 + 1 + 4 * (prefetch - 1) + (int(parts) << 7)


def _linear_workspace(dev: torch.device) -> torch.Tensor:
    """Launches of one stream run one after the other and share the partial-tile slots; launches of different streams (the two
    encoders of an SDXL edit) may overlap and get their own (32 MB each)."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    ws = _LINEAR_WS.get(key)
    if ws is None:
        if len(_LINEAR_WS) >= 8:
            _LINEAR_WS.clear()
        ws = _LINEAR_WS[key] = torch.zeros(int(load().emcid_linear_workspace_bytes()), dtype=torch.uint8, device=dev)
    return ws


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
           residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, cfg: int = -1) -> torch.Tensor:
    """act(x @ w.T + bias) + residual in one launch (csrc/gemm_f32.hip): x (M, K), w (N, K) like nn.Linear.weight, bias (N,),
    residual (M, N) row views; fp32, exact-f32 MFMA.  Launches of at most 128 tiles on a K of 2048 or more are cut by K over
    several workgroups per tile (partial tiles meet in a per-stream workspace, fixed summation order)."""
    if not linear_supported(x, w):
        raise EmcidHipError("linear: fp32 HBM operands with K contiguous, K % 16 == 0 and 16-byte aligned rows")
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    if out.shape != (M, N) or out.stride(1) != 1 or out.dtype != torch.float32:
        raise EmcidHipError("linear: out must be an (M, N) fp32 row view")
    if bias is not None and (bias.shape != (N,) or not bias.is_contiguous()):
        raise EmcidHipError("linear: bias must be a contiguous (N,) vector")
    if residual is not None and (residual.shape != (M, N) or residual.stride(1) != 1):
        raise EmcidHipError("linear: residual must be an (M, N) row view")
    if LINEAR_FLOPS["count"]:
        LINEAR_FLOPS["flops"] += 2.0 * M * N * K
        LINEAR_FLOPS["launches"] += 1
    ws = None
    if (cfg >= 128 or (cfg < 0 and K >= 2048 and -(-M // 128) * -(-N // 128) <= 128)) and not torch.cuda.is_current_stream_capturing():
        ws = _linear_workspace(x.device)
    try:
        _check(load().emcid_linear_ws_f32(_ptr(x, torch.float32, "x"), x.stride(0), _ptr(w, torch.float32, "w"), w.stride(0),
                                          _ptr(bias, torch.float32, "bias"), _ptr(residual, torch.float32, "residual"),
                                          residual.stride(0) if residual is not None else 0, _ptr(out, torch.float32, "out"),
                                          out.stride(0), M, N, K, int(act), int(cfg),
                                          C.c_void_p(ws.data_ptr()) if ws is not None else None, ws.numel() if ws is not None else 0,
                                          _stream(x)), "emcid_linear_ws_f32")
    except EmcidHipError:
        # the split-K form relies on its ticket counters being zero on entry; a launch that did not complete may have left
        # some behind — later launches on this stream would then never see a "last arriver".  Start from a fresh workspace.
        if ws is not None:
            _LINEAR_WS.pop((x.device.index if x.device.index is not None else torch.cuda.current_device(),
                            torch.cuda.current_stream(x.device).cuda_stream), None)
        raise
    return out


class SplitRows:
    """An fp32 matrix (rows, K) as two fp16 planes under per-row power-of-two scales (include/emcid_hip.h, "split fp16"):
    ``planes`` int32 (rows, K) — one 4-byte unit per element, [hi x 8][lo x 8] per group of 8 k —, ``inv_scale`` fp32 (rows,).
    ``f32``: the fp32 matrix itself when the producer wrote it too; ``bound``: for a weight, the device pair
    (largest row norm, largest |bias|) ``add_layernorm_sp`` turns into the scale of the projection's own split output."""
    __slots__ = ("planes", "inv_scale", "f32", "bound", "out_scale")

    def __init__(self, planes: torch.Tensor, inv_scale: torch.Tensor, f32: Optional[torch.Tensor] = None,
                 bound: Optional[torch.Tensor] = None, out_scale: Optional[torch.Tensor] = None):
        self.planes, self.inv_scale, self.f32, self.bound = planes, inv_scale, f32, bound
        self.out_scale = out_scale        # (2, rows): scale / inverse scale for the split OUTPUT of the projection that consumes this

    @property
    def shape(self):
        return self.planes.shape

    def index_select(self, idx: torch.Tensor) -> "SplitRows":
        return SplitRows(self.planes.index_select(0, idx), self.inv_scale.index_select(0, idx),
                         self.f32.index_select(0, idx) if self.f32 is not None else None)

    def rows(self, lo: int, hi: int) -> "SplitRows":
        return SplitRows(self.planes[lo:hi], self.inv_scale[lo:hi], self.f32[lo:hi] if self.f32 is not None else None)

    def float(self) -> torch.Tensor:
        """Back to fp32: the twin when there is one, else (hi + lo) * 2^-e (22-23 significant bits)."""
        if self.f32 is not None:
            return self.f32
        r, k = self.planes.shape
        h = self.planes.contiguous().view(torch.float16).view(r, k // 8, 2, 8).float()
        return ((h[:, :, 0] + h[:, :, 1]).reshape(r, k)) * self.inv_scale[:, None]


def split_supported(x: torch.Tensor) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] % 32 == 0 and x.stride(1) == 1
            and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and x.shape[0] > 0)


def split_rows(x: torch.Tensor, bias: Optional[torch.Tensor] = None, want_bound: bool = False) -> SplitRows:
    """fp32 rows -> split fp16 planes (csrc/gemm_sp16.hip: split_rows_kernel).  ``want_bound`` (weights): also the device pair
    (largest Euclidean row norm, largest |bias|)."""
    if not split_supported(x):
        raise EmcidHipError("split_rows: fp32 HBM rows with K contiguous, K % 32 == 0 and 16-byte aligned rows")
    rows, K = x.shape
    planes = torch.empty(rows, K, dtype=torch.int32, device=x.device)
    inv = torch.empty(rows, dtype=torch.float32, device=x.device)
    bound = None
    if want_bound:
        bound = torch.zeros(2, dtype=torch.float32, device=x.device)
        if bias is not None:
            bound[1:2] = bias.detach().abs().max()
    _check(load().emcid_split_rows_f16(_ptr(x, torch.float32, "x"), x.stride(0), rows, K, _ptr(planes), planes.stride(0), _ptr(inv),
                                       _ptr(bound), _stream(x)), "emcid_split_rows_f16")
    return SplitRows(planes, inv, None, bound)


def linear_sp(x: SplitRows, w: SplitRows, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
              residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, want_f32: bool = True,
              planes_scale: Optional[torch.Tensor] = None, cfg: int = -1):
    """act(x @ w.T + bias) + residual with both operands given as split fp16 matrices (three f16 MFMAs per k-step, fp32
    accumulate).  Returns the fp32 result, or — with ``planes_scale`` ((2, M): per-row 2^e bounding the result rows, and 2^-e) —
    the result as a SplitRows (its ``f32`` twin too unless ``want_f32=False``)."""
    M, K = x.planes.shape
    N = w.planes.shape[0]
    if w.planes.shape[1] != K or K % 32:
        raise EmcidHipError("linear_sp: operands (M, K) and (N, K) with K % 32 == 0")
    for t in (x.planes, w.planes):
        if t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16 or t.dtype != torch.int32:
            raise EmcidHipError("linear_sp: planes must be int32 row views with 16-byte aligned rows")
    dev = x.planes.device
    y = None
    if want_f32 or planes_scale is None:
        y = out if out is not None else torch.empty(M, N, dtype=torch.float32, device=dev)
        if y.shape != (M, N) or y.stride(1) != 1 or y.dtype != torch.float32:
            raise EmcidHipError("linear_sp: out must be an (M, N) fp32 row view")
    yp = None
    if planes_scale is not None:
        if N % 32 or planes_scale.shape != (2, M) or not planes_scale.is_contiguous():
            raise EmcidHipError("linear_sp: planes output needs N % 32 == 0 and a contiguous (2, M) scale / inverse scale")
        yp = torch.empty(M, N, dtype=torch.int32, device=dev)
    if bias is not None and (bias.shape != (N,) or not bias.is_contiguous()):
        raise EmcidHipError("linear_sp: bias must be a contiguous (N,) vector")
    if residual is not None and (residual.shape != (M, N) or residual.stride(1) != 1):
        raise EmcidHipError("linear_sp: residual must be an (M, N) row view")
    if LINEAR_FLOPS["count"]:
        LINEAR_FLOPS["flops"] += 2.0 * M * N * K
        LINEAR_FLOPS["launches"] += 1
    _check(load().emcid_linear_sp16_f32(
        _ptr(x.planes), x.planes.stride(0), _ptr(x.inv_scale, torch.float32, "x scale"), _ptr(w.planes), w.planes.stride(0),
        _ptr(w.inv_scale, torch.float32, "w scale"), _ptr(bias, torch.float32, "bias"), _ptr(residual, torch.float32, "residual"),
        residual.stride(0) if residual is not None else 0, _ptr(y), y.stride(0) if y is not None else 0, _ptr(yp),
        yp.stride(0) if yp is not None else 0, _ptr(planes_scale, torch.float32, "planes scale"), M, N, K, int(act), int(cfg),
        _stream(x.planes)), "emcid_linear_sp16_f32")
    if planes_scale is None:
        return y
    return SplitRows(yp, planes_scale[1], y)


def add_layernorm(a: torch.Tensor, b: Optional[torch.Tensor], ln: torch.nn.LayerNorm, want_sum: bool = True):
    """(a + b, LayerNorm(a + b)) in one pass; a, b (rows, cols) fp32 with unit column stride.  ``b=None``: LayerNorm(a)
    alone (returns (a, LayerNorm(a)); nothing but z is written)."""
    rows, cols = a.shape
    if a.stride(1) != 1 or (b is not None and (b.stride(1) != 1 or b.shape != a.shape)):
        raise EmcidHipError("add_layernorm: (rows, cols) operands with unit column stride")
    if ln.weight is None or ln.bias is None or tuple(ln.normalized_shape) != (cols,):
        raise EmcidHipError("add_layernorm: LayerNorm over the last dimension with affine parameters")
    y = torch.empty(rows, cols, dtype=torch.float32, device=a.device) if (b is not None and want_sum) else None
    z = torch.empty(rows, cols, dtype=torch.float32, device=a.device)
    _check(load().emcid_add_layernorm_f32(_ptr(a, torch.float32, "a"), a.stride(0), _ptr(b, torch.float32, "b"),
                                          b.stride(0) if b is not None else 0,
                                          _ptr(ln.weight, torch.float32, "gamma"), _ptr(ln.bias, torch.float32, "beta"),
                                          float(ln.eps), rows, cols, _ptr(y), _ptr(z), _stream(a)), "emcid_add_layernorm_f32")
    return (a if b is None else y), z


def add_layernorm_sp(a: torch.Tensor, b: Optional[torch.Tensor], ln: torch.nn.LayerNorm, want_sum: bool = True,
                     want_f32: bool = False, bound: Optional[torch.Tensor] = None):
    """``add_layernorm`` with the LayerNorm's output as a split-fp16 matrix for the projection that consumes it:
    returns (a + b | a, SplitRows of LayerNorm(a + b) [with .f32 when ``want_f32``]).  ``bound`` (the consuming weight's
    ``SplitRows.bound``): the result's ``out_scale`` (2, rows) = the per-row 2^e (and 2^-e) under which that projection may write its
    own output as planes (``linear_sp(planes_scale=...)``)."""
    rows, cols = a.shape
    if a.stride(1) != 1 or (b is not None and (b.stride(1) != 1 or b.shape != a.shape)):
        raise EmcidHipError("add_layernorm_sp: (rows, cols) operands with unit column stride")
    if ln.weight is None or ln.bias is None or tuple(ln.normalized_shape) != (cols,) or cols % 32 or cols > 2048:
        raise EmcidHipError("add_layernorm_sp: LayerNorm over the last dimension with affine parameters, cols % 32 == 0, <= 2048")
    dev = a.device
    y = torch.empty(rows, cols, dtype=torch.float32, device=dev) if (b is not None and want_sum) else None
    z = torch.empty(rows, cols, dtype=torch.float32, device=dev) if want_f32 else None
    planes = torch.empty(rows, cols, dtype=torch.int32, device=dev)
    inv = torch.empty(rows, dtype=torch.float32, device=dev)
    out_scale = torch.empty(2, rows, dtype=torch.float32, device=dev) if bound is not None else None
    _check(load().emcid_add_layernorm_sp16(_ptr(a, torch.float32, "a"), a.stride(0), _ptr(b, torch.float32, "b"),
                                           b.stride(0) if b is not None else 0,
                                           _ptr(ln.weight, torch.float32, "gamma"), _ptr(ln.bias, torch.float32, "beta"),
                                           float(ln.eps), rows, cols, _ptr(y), _ptr(z), _ptr(planes), planes.stride(0), _ptr(inv),
                                           _ptr(bound, torch.float32, "bound"), _ptr(out_scale), _stream(a)),
           "emcid_add_layernorm_sp16")
    return (a if b is None else y), SplitRows(planes, inv, z, None, out_scale)


def embed_layernorm_sp(tok_emb: torch.Tensor, pos_emb: torch.Tensor, token: torch.Tensor, position: torch.Tensor,
                       ln: torch.nn.LayerNorm):
    """``embed_layernorm`` with the LayerNorm's output as a split-fp16 matrix: (embeddings, SplitRows)."""
    rows, cols = token.numel(), tok_emb.shape[1]
    if tok_emb.stride(1) != 1 or pos_emb.stride(1) != 1 or pos_emb.shape[1] != cols or position.numel() != rows:
        raise EmcidHipError("embed_layernorm_sp: embedding tables with unit column stride and one index pair per row")
    if ln.weight is None or ln.bias is None or tuple(ln.normalized_shape) != (cols,) or cols % 32 or cols > 2048:
        raise EmcidHipError("embed_layernorm_sp: LayerNorm over the last dimension with affine parameters, cols % 32 == 0, <= 2048")
    y = torch.empty(rows, cols, dtype=torch.float32, device=tok_emb.device)
    planes = torch.empty(rows, cols, dtype=torch.int32, device=tok_emb.device)
    inv = torch.empty(rows, dtype=torch.float32, device=tok_emb.device)
    _check(load().emcid_embed_layernorm_sp16(_ptr(tok_emb, torch.float32, "tok_emb"), tok_emb.stride(0), tok_emb.shape[0],
                                             _ptr(pos_emb, torch.float32, "pos_emb"), pos_emb.stride(0), pos_emb.shape[0],
                                             _ptr(token, torch.int64, "token"), _ptr(position, torch.int32, "position"),
                                             _ptr(ln.weight, torch.float32, "gamma"), _ptr(ln.bias, torch.float32, "beta"),
                                             float(ln.eps), rows, cols, _ptr(y), None, _ptr(planes), planes.stride(0), _ptr(inv),
                                             _stream(tok_emb)), "emcid_embed_layernorm_sp16")
    return y, SplitRows(planes, inv)


def tree_attention_sp_supported(anc: torch.Tensor, H: int, D: int) -> bool:
    return bool(load().emcid_tree_attention_sp16_supported(anc.shape[1], H, D)) and (H * D) % 32 == 0


def tree_attention_sp(q, k, v, anc, depth, H: int, scale=None, rows=None) -> SplitRows:
    """``tree_attention`` with the result as a split-fp16 matrix for the out-projection (short chains only)."""
    U, HD = k.shape
    D = HD // H
    for t in (q, k, v):
        if t.stride(1) != 1 or t.shape[1] != HD:
            raise EmcidHipError("tree_attention_sp: q/k/v must be row-major")
    if k.stride(0) != v.stride(0):
        raise EmcidHipError("tree_attention_sp: k and v must share a leading dimension")
    n = U if rows is None else rows.numel()
    if q.shape[0] != n:
        raise EmcidHipError(f"tree_attention_sp: {q.shape[0]} query rows for {n} query nodes")
    planes = torch.empty(n, HD, dtype=torch.int32, device=q.device)
    inv = torch.empty(n, dtype=torch.float32, device=q.device)
    scale = float(D ** -0.5 if scale is None else scale)
    _check(load().emcid_tree_attention_sp16(
        _ptr(q, torch.float32, "q"), q.stride(0), _ptr(k, torch.float32, "k"), _ptr(v, torch.float32, "v"), k.stride(0),
        _ptr(anc, torch.int32, "anc"), anc.stride(0), _ptr(depth, torch.int32, "depth"),
        _ptr(rows, torch.int32, "rows") if rows is not None else None, n, H, D, scale, _ptr(planes), planes.stride(0), _ptr(inv),
        _stream(q)), "emcid_tree_attention_sp16")
    return SplitRows(planes, inv)


def embed_layernorm(tok_emb: torch.Tensor, pos_emb: torch.Tensor, token: torch.Tensor, position: torch.Tensor,
                    ln: torch.nn.LayerNorm):
    """(tok_emb[token] + pos_emb[position], LayerNorm of it) in one launch; token int64 (rows,), position int32 (rows,)."""
    rows, cols = token.numel(), tok_emb.shape[1]
    if tok_emb.stride(1) != 1 or pos_emb.stride(1) != 1 or pos_emb.shape[1] != cols or position.numel() != rows:
        raise EmcidHipError("embed_layernorm: embedding tables with unit column stride and one index pair per row")
    if ln.weight is None or ln.bias is None or tuple(ln.normalized_shape) != (cols,):
        raise EmcidHipError("embed_layernorm: LayerNorm over the last dimension with affine parameters")
    y = torch.empty(rows, cols, dtype=torch.float32, device=tok_emb.device)
    z = torch.empty_like(y)
    _check(load().emcid_embed_layernorm_f32(_ptr(tok_emb, torch.float32, "tok_emb"), tok_emb.stride(0), tok_emb.shape[0],
                                            _ptr(pos_emb, torch.float32, "pos_emb"), pos_emb.stride(0), pos_emb.shape[0],
                                            _ptr(token, torch.int64, "token"), _ptr(position, torch.int32, "position"),
                                            _ptr(ln.weight, torch.float32, "gamma"), _ptr(ln.bias, torch.float32, "beta"),
                                            float(ln.eps), rows, cols, _ptr(y), _ptr(z), _stream(tok_emb)),
           "emcid_embed_layernorm_f32")
    return y, z


def tree_attention(q, k, v, anc, depth, H: int, scale=None, rows=None):
    """k/v (U, H*D) fp32 row views sharing one leading dimension; q (n, H*D) in query order; anc (U, S) int32;
    depth (U,) int32; rows: optional (n,) int32 query nodes (None: all U nodes, q has U rows).  Returns (n, H*D)."""
    U, HD = k.shape
    D = HD // H
    for t in (q, k, v):
        if t.stride(1) != 1 or t.shape[1] != HD:
            raise EmcidHipError("tree_attention: q/k/v must be row-major")
    if k.stride(0) != v.stride(0):
        raise EmcidHipError("tree_attention: k and v must share a leading dimension")
    n = U if rows is None else rows.numel()
    if q.shape[0] != n:
        raise EmcidHipError(f"tree_attention: {q.shape[0]} query rows for {n} query nodes")
    out = torch.empty(n, HD, dtype=torch.float32, device=q.device)
    scale = float(D ** -0.5 if scale is None else scale)
    _check(load().emcid_tree_attention_f32(
        _ptr(q, torch.float32, "q"), q.stride(0), _ptr(k, torch.float32, "k"), _ptr(v, torch.float32, "v"), k.stride(0),
        _ptr(anc, torch.int32, "anc"), anc.stride(0), _ptr(depth, torch.int32, "depth"),
        _ptr(rows, torch.int32, "rows") if rows is not None else None, n, H, D, scale, _ptr(out), out.stride(0),
        _stream(q)), "emcid_tree_attention_f32")
    return out


# ---- native layer runner of the trie forward (csrc/clip_layers.hip) -----------------------------------------------------------

class ClipLayerSp16(C.Structure):
    """include/emcid_hip.h: emcid_clip_layer_sp16."""
    _fields_ = [(n, C.c_void_p) for n in (
        "ln1_gamma", "ln1_beta", "ln2_gamma", "ln2_beta", "qkv_planes", "qkv_inv_scale", "qkv_bias", "out_planes",
        "out_inv_scale", "out_bias", "fc1_planes", "fc1_inv_scale", "fc1_bias", "fc1_bound", "fc2_planes", "fc2_inv_scale",
        "fc2_bias")] + [("ln1_eps", C.c_float), ("ln2_eps", C.c_float), ("act", C.c_int32), ("reserved", C.c_int32)]


_CLIP_WS = {}       # (device index, stream handle, rows, h, d) -> workspace of the layer runner


def clip_workspace(dev: torch.device, rows: int, h: int, d: int) -> torch.Tensor:
    """Launches of one stream run one after the other and may share the runner's scratch buffers."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream,
           rows, h, d)
    ws = _CLIP_WS.get(key)
    if ws is None:
        if len(_CLIP_WS) >= 16:
            _CLIP_WS.clear()
        ws = _CLIP_WS[key] = torch.empty(int(load().emcid_clip_workspace_bytes(rows, h, d)), dtype=torch.uint8, device=dev)
    return ws


def clip_layers(layer_array, first: int, count: int, rows: int, h: int, d: int, heads: int, scale: float, anc: torch.Tensor,
                depth: torch.Tensor, hs: torch.Tensor, x: "SplitRows", next_ln: Optional[torch.nn.LayerNorm]):
    """``count`` whole layers starting at ``layer_array[first]`` on every trie node, in place on ``hs`` (rows, h) and on the
    planes of ``x`` (LN1 of hs in, the next layer's LN1 of the result out when ``next_ln`` is given)."""
    ws = clip_workspace(hs.device, rows, h, d)
    if LINEAR_FLOPS["count"]:
        LINEAR_FLOPS["flops"] += count * 2.0 * rows * h * (4 * h + 2 * d)
        LINEAR_FLOPS["launches"] += 4 * count
    base = C.addressof(layer_array) + first * C.sizeof(ClipLayerSp16)
    g = _ptr(next_ln.weight, torch.float32, "gamma") if next_ln is not None else None
    b = _ptr(next_ln.bias, torch.float32, "beta") if next_ln is not None else None
    _check(load().emcid_clip_layers_sp16(C.c_void_p(base), count, rows, h, d, heads, float(scale), _ptr(anc, torch.int32, "anc"),
                                         anc.stride(0), _ptr(depth, torch.int32, "depth"), _ptr(hs, torch.float32, "hs"),
                                         _ptr(x.planes), _ptr(x.inv_scale), g, b, float(next_ln.eps) if next_ln is not None else 0.0,
                                         C.c_void_p(ws.data_ptr()), ws.numel(), _stream(hs)), "emcid_clip_layers_sp16")


def clip_layer_head(layer_array, index: int, rows: int, h: int, d: int, heads: int, scale: float, anc: torch.Tensor,
                    depth: torch.Tensor, rows_sel: Optional[torch.Tensor], hs: torch.Tensor, x: "SplitRows", want_f32: bool):
    """Attention block + fc1 of layer ``layer_array[index]``: returns (mid (n, h), SplitRows of fc2's input (n, d) [with its
    fp32 twin when ``want_f32``]); n = every node, or the ``rows_sel`` query nodes."""
    dev = hs.device
    n = rows if rows_sel is None else rows_sel.numel()
    ws = clip_workspace(dev, rows, h, d)
    mid = torch.empty(n, h, dtype=torch.float32, device=dev)
    f_planes = torch.empty(n, d, dtype=torch.int32, device=dev)
    f_scale = torch.empty(2, n, dtype=torch.float32, device=dev)
    f_f32 = torch.empty(n, d, dtype=torch.float32, device=dev) if want_f32 else None
    if LINEAR_FLOPS["count"]:
        LINEAR_FLOPS["flops"] += 2.0 * h * (rows * (3 * h if rows_sel is None else 2 * h) + n * (h if rows_sel is not None else 0)
                                            + n * h + n * d)
        LINEAR_FLOPS["launches"] += 3 if rows_sel is None else 4
    base = C.addressof(layer_array) + index * C.sizeof(ClipLayerSp16)
    _check(load().emcid_clip_layer_head_sp16(
        C.c_void_p(base), rows, h, d, heads, float(scale), _ptr(anc, torch.int32, "anc"), anc.stride(0),
        _ptr(depth, torch.int32, "depth"), _ptr(rows_sel, torch.int32, "rows") if rows_sel is not None else None, n,
        _ptr(hs, torch.float32, "hs"), _ptr(x.planes), _ptr(x.inv_scale), _ptr(mid), _ptr(f_planes), _ptr(f_scale), _ptr(f_f32),
        C.c_void_p(ws.data_ptr()), ws.numel(), _stream(hs)), "emcid_clip_layer_head_sp16")
    return mid, SplitRows(f_planes, f_scale[1], f_f32)


def clip_layer_tail(layer_array, index: int, h: int, d: int, f: "SplitRows", mid: torch.Tensor,
                    next_ln: Optional[torch.nn.LayerNorm]):
    """fc2 + residual add of layer ``layer_array[index]`` and the next layer's LN1 as planes: (hs (n, h), SplitRows | None)."""
    dev = mid.device
    n = mid.shape[0]
    hs = torch.empty(n, h, dtype=torch.float32, device=dev)
    x = None
    if next_ln is not None:
        x = SplitRows(torch.empty(n, h, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.float32, device=dev))
    if LINEAR_FLOPS["count"]:
        LINEAR_FLOPS["flops"] += 2.0 * n * h * d
        LINEAR_FLOPS["launches"] += 1
    base = C.addressof(layer_array) + index * C.sizeof(ClipLayerSp16)
    _check(load().emcid_clip_layer_tail_sp16(
        C.c_void_p(base), n, h, d, _ptr(f.planes), _ptr(f.inv_scale), _ptr(mid, torch.float32, "mid"), _ptr(hs),
        _ptr(next_ln.weight, torch.float32, "gamma") if next_ln is not None else None,
        _ptr(next_ln.bias, torch.float32, "beta") if next_ln is not None else None,
        float(next_ln.eps) if next_ln is not None else 0.0, _ptr(x.planes) if x is not None else None,
        _ptr(x.inv_scale) if x is not None else None, _stream(mid)), "emcid_clip_layer_tail_sp16")
    return hs, x


def clip_edit_layer_tail(layer_array, index: int, h: int, d: int, f: "SplitRows", mid: torch.Tensor, lookup: torch.Tensor,
                         seg: torch.Tensor, zs_t: torch.Tensor, factors: "CovFactors", layer_index: int, edit_weight: float,
                         layers_left: int, W0: torch.Tensor, W: torch.Tensor, ws: "DualWorkspace", lam: Optional[float],
                         next_ln: Optional[torch.nn.LayerNorm], last: bool, want_dw: bool = True, zc_split: bool = True):
    """The rest of an edited layer behind ``clip_layer_head`` in ONE C call: keys, Zc, the apply-only dual solve (W = W0 + dW), the
    new weight's planes and — unless ``last`` — fc2 + residual + the next layer's LN1.  Returns dict(K, Zc, dW, hs, x)."""
    dev = mid.device
    n = f.f32.shape[0]
    N = seg.numel() - 1
    K = torch.empty(N, d, dtype=torch.float32, device=dev)
    Zc = torch.empty(N, h, dtype=torch.float32, device=dev)
    dW = torch.empty(h, d, dtype=torch.float32, device=dev) if want_dw else None
    # Zc = fc2(keys) on the split-fp16 kernel against the layer's own fc2 planes (scratch for the keys' planes): the caller
    # says so with zc_split — the planes in ``layer_array`` must then be those of W as it is now
    kp = torch.empty(N, d, dtype=torch.int32, device=dev) if zc_split else None
    ks = torch.empty(N, dtype=torch.float32, device=dev) if zc_split else None
    hs = x = None
    if not last:
        hs = torch.empty(n, h, dtype=torch.float32, device=dev)
        if next_ln is not None:
            x = SplitRows(torch.empty(n, h, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.float32, device=dev))
    lws = _linear_workspace(dev) if (d >= 2048 and -(-N // 128) * -(-h // 128) <= 128
                                     and not torch.cuda.is_current_stream_capturing()) else None
    if LINEAR_FLOPS["count"]:
        LINEAR_FLOPS["flops"] += 2.0 * N * h * d + (0.0 if last else 2.0 * n * h * d)
        LINEAR_FLOPS["launches"] += 1 if last else 2
    base = C.addressof(layer_array) + index * C.sizeof(ClipLayerSp16)
    use_inverse = layer_index in factors.have_inverse
    _check(load().emcid_clip_edit_layer_tail_sp16(
        C.c_void_p(base), n, h, d, _ptr(f.f32, torch.float32, "fc2 input"), _ptr(f.planes), _ptr(f.inv_scale),
        _ptr(mid, torch.float32, "mid"), _ptr(lookup, torch.int64, "lookup"), _ptr(seg, torch.int64, "seg"), lookup.numel(), N,
        _ptr(zs_t, torch.float32, "zs_t"), float(edit_weight), int(layers_left), factors.lam_ratio(lam), _ptr(factors.buf),
        factors.n_layers, int(layer_index), int(bool(use_inverse)), _ptr(W0, torch.float32, "W0"), _ptr(W, torch.float32, "W"),
        _ptr(dW), _ptr(K), _ptr(Zc), _ptr(kp), _ptr(ks), _ptr(ws.buf), ws.nbytes, _ptr(ws.info, torch.int32),
        C.c_void_p(lws.data_ptr()) if lws is not None else None, lws.numel() if lws is not None else 0, _ptr(hs),
        _ptr(next_ln.weight, torch.float32, "gamma") if (next_ln is not None and not last) else None,
        _ptr(next_ln.bias, torch.float32, "beta") if (next_ln is not None and not last) else None,
        float(next_ln.eps) if (next_ln is not None and not last) else 0.0, _ptr(x.planes) if x is not None else None,
        _ptr(x.inv_scale) if x is not None else None, _stream(mid)), "emcid_clip_edit_layer_tail_sp16")
    return {"K": K, "Zc": Zc, "dW": dW, "hs": hs, "x": x}


# ---- dual (Woodbury) solver ---------------------------------------------------------------------------------------------

class CovFactors:
    """Cholesky factors of M_l = lam * C'_l for every edited layer (batched factorization, one workspace)."""

    def __init__(self, n_layers: int, d: int, device):
        self.n_layers, self.d = n_layers, d
        self.nbytes = int(load().emcid_cov_factor_workspace_bytes(n_layers, d))
        self.buf = torch.zeros(self.nbytes // 8, dtype=torch.float64, device=device)
        self.info = torch.zeros(1, dtype=torch.int32, device=device)
        self.dp = (d + NB - 1) // NB * NB
        self._inv = ((self.dp + 511) // 512) * (512 * 512 + 256 * 256)   # csrc/common.h inv_doubles
        self.have_inverse = set()                                        # layers whose X = inv(L) has been built
        self.ready = None       # HIP event after the last kernel that wrote this workspace on another stream (or None)
        self.cached = False     # True once the edit engine shares it between edits (then it is read-only)
        self.lam = None         # the lam the workspace was factored with (factor_cov); edits with another lam pass the ratio
        self.edit_weight = None

    def lam_ratio(self, lam: Optional[float]) -> float:
        """lam of an edit / the lam of the factorization: the gain the dual stages fold into Kt64 and Rt
        (include/emcid_hip.h, "lam_ratio")."""
        if lam is None or self.lam is None or float(lam) == self.lam:
            return 1.0
        if not (lam > 0.0 and self.lam > 0.0):
            raise EmcidHipError(f"mom2_update_weight must be positive for the Cholesky path (got {lam})")
        return float(lam) / float(self.lam)

    def L(self, layer: int) -> torch.Tensor:
        """(dp, dp) view of the Cholesky factor of lam*C'_layer (lower triangle valid)."""
        o = (self.n_layers + layer) * self.dp * self.dp
        return self.buf[o:o + self.dp * self.dp].view(self.dp, self.dp)

    def X(self, layer: int) -> torch.Tensor:
        """(dp, dp) view of inv(L_layer), explicit (lower triangle valid; nothing above it is ever read)."""
        o = self.n_layers * (2 * self.dp * self.dp + self._inv) + layer * self.dp * self.dp
        return self.buf[o:o + self.dp * self.dp].view(self.dp, self.dp)


def cov_inverse(factors: "CovFactors", first: int = 0, count: Optional[int] = None):
    """Build X = inv(L) for layers [first, first + count) of a factored workspace (default: all), batched.
    Asynchronous on the current stream; needs factor_cov earlier on it."""
    count = factors.n_layers - first if count is None else count
    _check(load().emcid_cov_inverse_f64(_ptr(factors.buf), factors.n_layers, factors.d, int(first), int(count),
                                        _stream(factors.buf)), "emcid_cov_inverse_f64")
    factors.have_inverse.update(range(first, first + count))


def factor_cov(covs, lam: float, edit_weight: float, factors: Optional[CovFactors] = None, inverse: bool = True) -> CovFactors:
    """covs: list of (d, d) fp32 contiguous HBM tensors (one per edited layer, forward order).  Asynchronous on the
    current stream.  ``inverse=False`` leaves the explicit inverse factors to per-layer ``cov_inverse`` calls."""
    d = covs[0].shape[0]
    for c in covs:
        assert c.shape == (d, d) and c.is_contiguous()
    if factors is None or factors.n_layers != len(covs) or factors.d != d:
        factors = CovFactors(len(covs), d, covs[0].device)
    arr = (C.c_void_p * len(covs))(*[_ptr(c, torch.float32, "C").value for c in covs])
    _check(load().emcid_factor_cov_f64(arr, len(covs), d, float(lam), float(edit_weight), _ptr(factors.buf), factors.nbytes,
                                       _ptr(factors.info, torch.int32), _stream(covs[0])), "emcid_factor_cov_f64")
    factors.have_inverse = set()
    factors.lam, factors.edit_weight = float(lam), float(edit_weight)
    if inverse:
        cov_inverse(factors)
    return factors


class DualWorkspace:
    def __init__(self, N: int, d: int, h: int, device):
        self.key = (N, d, h)
        self.nbytes = int(load().emcid_edit_dual_workspace_bytes(N, d, h))
        self.buf = torch.zeros(self.nbytes // 8, dtype=torch.float64, device=device)     # zero: the stream-K ticket counters
        self.info = torch.zeros(1, dtype=torch.int32, device=device)
        self.Np, self.dp = (N + NB - 1) // NB * NB, (d + NB - 1) // NB * NB
        pt = load().emcid_edit_dual_pt(_ptr(self.buf), N, d, h)
        off = (pt - self.buf.data_ptr()) // 8
        self.Pt = self.buf[off:off + self.Np * self.dp].view(self.Np, self.dp)   # the Pt stack, rows = concepts
        yt = load().emcid_edit_dual_yt(_ptr(self.buf), N, d, h)
        off = (yt - self.buf.data_ptr()) // 8
        self.Yt = self.buf[off:off + self.Np * self.dp].view(self.Np, self.dp)   # the Yt stack of the apply-only form
        hp = h + (h % 2)
        off = (load().emcid_edit_dual_s(_ptr(self.buf), N, d, h) - self.buf.data_ptr()) // 8
        self.S = self.buf[off:off + self.Np * self.Np].view(self.Np, self.Np)    # N x N system (column-sharded form: partial sums)
        off = (load().emcid_edit_dual_u(_ptr(self.buf), N, d, h) - self.buf.data_ptr()) // 8
        self.U = self.buf[off:off + hp * self.dp].view(hp, self.dp)[:h]          # U (h, dp) f64 partial sums of the same form


def edit_layer_dual(K, Zc, zs_t, factors: CovFactors, layer_index: int, edit_weight: float, layers_left: int,
                    W0=None, W=None, want_factors: bool = False, want_dw: bool = True, ws: Optional[DualWorkspace] = None,
                    rows=None, gather_pt=None, use_inverse: Optional[bool] = None, lam: Optional[float] = None):
    """One edited layer through the dual solver.  ``lam``: this edit's mom2_update_weight when it differs from the one
    ``factors`` was built with (None: the same).  ``rows=(lo, hi)`` + ``gather_pt(Pt_rows) -> all rows`` split the
    M-solves over ranks.  ``use_inverse``: solve against M with GEMMs on X = inv(L) (default: if cov_inverse built it)
    or by block substitution with L.  Returns dict(adj_k (d,N) | None, Rt (N,h) | None, dW, ws)."""
    if use_inverse is None:
        use_inverse = layer_index in factors.have_inverse
    N, d = K.shape
    h = Zc.shape[1]
    for t, nm in ((K, "K"), (Zc, "Zc"), (zs_t, "zs_t")):
        assert t.is_contiguous(), nm
    assert zs_t.shape == (N, h) and factors.d == d
    if ws is None or ws.key != (N, d, h):
        ws = DualWorkspace(N, d, h, K.device)
    lo, hi = rows if rows is not None else (0, N)
    lib = load()
    _check(lib.emcid_edit_dual_stage1_f64(
        _ptr(K, torch.float32, "K"), _ptr(Zc, torch.float32, "Zc"), _ptr(zs_t, torch.float32, "zs_t"), N, d, h,
        float(edit_weight), int(layers_left), factors.lam_ratio(lam), _ptr(factors.buf), factors.n_layers, int(layer_index), lo, hi,
        int(bool(use_inverse)), _ptr(ws.buf), ws.nbytes, _stream(K)), "emcid_edit_dual_stage1_f64")
    if gather_pt is not None:
        ws.Pt[:N].copy_(gather_pt(ws.Pt[lo:hi]))
    dev = K.device
    adj_k = torch.empty(d, N, dtype=torch.float64, device=dev) if want_factors else None
    Rt = torch.empty(N, h, dtype=torch.float64, device=dev) if want_factors else None
    dW = torch.empty(h, d, dtype=torch.float32, device=dev) if want_dw else None
    if W is not None:
        assert W.is_contiguous() and W.shape == (h, d) and W0 is not None and W0.is_contiguous()
    _check(lib.emcid_edit_dual_stage2_f64(N, d, h, factors.lam_ratio(lam), _ptr(W0, torch.float32, "W0"), _ptr(W, torch.float32, "W"), _ptr(adj_k),
                                          _ptr(Rt), _ptr(dW), _ptr(ws.buf), ws.nbytes, _ptr(ws.info, torch.int32),
                                          _stream(K)), "emcid_edit_dual_stage2_f64")
    return {"adj_k": adj_k, "Rt": Rt, "dW": dW, "ws": ws}


def column_tiles(rank: int, world: int, n_tiles: int):
    """This rank's 128-wide column tiles of the d dimension for the column-sharded solve.  Tile t of the triangular factor
    costs ~(t + 1) (its contraction depth), so the tiles are dealt heaviest first, each to the least loaded rank so far
    (ties: lowest rank) — the same table on every rank, per-rank cost within a few percent of the mean (24 tiles over 8
    ranks: 36..39 of a mean 37.5)."""
    load_ = [0] * world
    owner = {}
    for t in range(n_tiles - 1, -1, -1):
        r = min(range(world), key=lambda i: (load_[i], i))
        owner[t] = r
        load_[r] += t + 1
    return [t for t in range(n_tiles) if owner[t] == rank]


class _HipColsBackend:
    """The two stages of the column-sharded solve on the C library (include/emcid_hip.h)."""

    def __init__(self, K, Zc, zs_t, factors, layer_index, edit_weight, layers_left, ws, lam=None):
        self.a = (K, Zc, zs_t, factors, layer_index, edit_weight, layers_left, ws)
        self.lam_ratio = factors.lam_ratio(lam)

    def stage1(self, tiles):
        K, Zc, zs_t, factors, layer_index, edit_weight, layers_left, ws = self.a
        N, d = K.shape
        arr = (C.c_int * len(tiles))(*tiles)
        _check(load().emcid_edit_dual_cols_stage1_f64(
            _ptr(K, torch.float32, "K"), _ptr(Zc, torch.float32, "Zc"), _ptr(zs_t, torch.float32, "zs_t"), N, d, Zc.shape[1],
            float(edit_weight), int(layers_left), self.lam_ratio, _ptr(factors.buf), factors.n_layers, int(layer_index), arr,
            len(tiles), _ptr(ws.buf), ws.nbytes, _stream(K)), "emcid_edit_dual_cols_stage1_f64")
        return ws.S

    def stage2(self, tiles):
        K, Zc, zs_t, factors, layer_index, edit_weight, layers_left, ws = self.a
        N, d = K.shape
        arr = (C.c_int * len(tiles))(*tiles)
        _check(load().emcid_edit_dual_cols_stage2_f64(N, d, Zc.shape[1], _ptr(factors.buf), factors.n_layers, int(layer_index), arr,
                                                      len(tiles), _ptr(ws.buf), ws.nbytes, _ptr(ws.info, torch.int32), _stream(K)),
               "emcid_edit_dual_cols_stage2_f64")
        return ws.U

    def apply(self, U, W0, W, want_dw):
        K, Zc = self.a[0], self.a[1]
        h, d = Zc.shape[1], K.shape[1]
        dW = torch.empty(h, d, dtype=torch.float32, device=K.device) if want_dw else None
        _check(load().emcid_apply_update2d_f32(_ptr(U), U.stride(0), _ptr(W0, torch.float32, "W0"), _ptr(W, torch.float32, "W"),
                                               _ptr(dW), h, d, _stream(K)), "emcid_apply_update2d_f32")
        return dW


def edit_layer_dual_cols(K, Zc, zs_t, factors, layer_index: int, edit_weight: float, layers_left: int, W0, W,
                         tiles, all_reduce, want_dw: bool = True, ws: Optional["DualWorkspace"] = None, backend=None,
                         lam: Optional[float] = None):
    """Apply-only dual solver with the layer's GEMMs split over ranks by column tiles of d (include/emcid_hip.h,
    "COLUMN-SHARDED").  ``tiles``: this rank's tile indices (column_tiles); ``all_reduce(t)``: sums a tensor over the
    ranks in place (two calls: the N x N partial S, the h x dp partial U).  Needs the layer's explicit inverse factor.
    ``backend``: object with stage1/stage2/apply (default: the C library; tests/test_dist_cpu.py passes a torch-CPU
    restatement of the two stages to run this very skeleton under gloo)."""
    N, d = K.shape
    h = Zc.shape[1]
    tiles = sorted(int(t) for t in tiles)
    if not tiles:
        raise EmcidHipError("a rank without a column tile: world size exceeds d / 128")
    if backend is None:
        for t, nm in ((K, "K"), (Zc, "Zc"), (zs_t, "zs_t"), (W, "W"), (W0, "W0")):
            assert t.is_contiguous(), nm
        assert zs_t.shape == (N, h) and factors.d == d and W.shape == (h, d)
        if layer_index not in factors.have_inverse:
            raise EmcidHipError("edit_layer_dual_cols needs the explicit inverse factor of the layer (cov_inverse)")
        if ws is None or ws.key != (N, d, h):
            ws = DualWorkspace(N, d, h, K.device)
        backend = _HipColsBackend(K, Zc, zs_t, factors, layer_index, edit_weight, layers_left, ws, lam)
    S = backend.stage1(tiles)            # partial S_r = Yc Yc^T (no identity)
    all_reduce(S)
    U = backend.stage2(tiles)            # partial U_r = (Z^T Yc) X[tiles, :]
    all_reduce(U)
    dW = backend.apply(U, W0, W, want_dw)
    return {"dW": dW, "ws": ws}


def edit_layer_dual_apply(K, Zc, zs_t, factors: CovFactors, layer_index: int, edit_weight: float, layers_left: int,
                          W0, W, want_dw: bool = True, ws: Optional[DualWorkspace] = None, rows=None, gather_yt=None,
                          use_inverse: Optional[bool] = None, on_factor_start=None, lam: Optional[float] = None):
    """Apply-only dual solver: W = W0 + float(U) without ever forming adj_k.  ``on_factor_start()``: called (host side)
    right after S = I + Yt Yt^T has been enqueued, i.e. the stream position where the latency-bound Cholesky of S
    begins.  Returns dict(dW, ws)."""
    if use_inverse is None:
        use_inverse = layer_index in factors.have_inverse
    N, d = K.shape
    h = Zc.shape[1]
    for t, nm in ((K, "K"), (Zc, "Zc"), (zs_t, "zs_t"), (W, "W"), (W0, "W0")):
        assert t.is_contiguous(), nm
    assert zs_t.shape == (N, h) and factors.d == d and W.shape == (h, d)
    if ws is None or ws.key != (N, d, h):
        ws = DualWorkspace(N, d, h, K.device)
    lo, hi = rows if rows is not None else (0, N)
    lib = load()
    _check(lib.emcid_edit_dual_apply_stage1_f64(
        _ptr(K, torch.float32, "K"), _ptr(Zc, torch.float32, "Zc"), _ptr(zs_t, torch.float32, "zs_t"), N, d, h,
        float(edit_weight), int(layers_left), factors.lam_ratio(lam), _ptr(factors.buf), factors.n_layers, int(layer_index), lo, hi,
        int(bool(use_inverse)), _ptr(ws.buf), ws.nbytes, _stream(K)), "emcid_edit_dual_apply_stage1_f64")
    if gather_yt is not None:
        ws.Yt[:N].copy_(gather_yt(ws.Yt[lo:hi]))
    dW = torch.empty(h, d, dtype=torch.float32, device=K.device) if want_dw else None
    if on_factor_start is not None:
        _check(lib.emcid_edit_dual_apply_assemble_f64(N, d, h, _ptr(ws.buf), ws.nbytes, _stream(K)),
               "emcid_edit_dual_apply_assemble_f64")
        on_factor_start()
    _check(lib.emcid_edit_dual_apply_stage2_f64(N, d, h, _ptr(factors.buf), factors.n_layers, int(layer_index),
                                                int(bool(use_inverse)), int(on_factor_start is not None),
                                                _ptr(W0, torch.float32, "W0"), _ptr(W, torch.float32, "W"), _ptr(dW),
                                                _ptr(ws.buf), ws.nbytes, _ptr(ws.info, torch.int32), _stream(K)),
           "emcid_edit_dual_apply_stage2_f64")
    return {"dW": dW, "ws": ws}
