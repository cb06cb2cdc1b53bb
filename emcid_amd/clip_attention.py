"""Routes a HF CLIP text encoder's attention through the fused HIP kernel (csrc/attention.hip) for the
duration of an edit / statistics forward.

transformers >= 4.48 looks attention up by name (``config._attn_implementation``) in a registry; we register
``emcid_hip`` there (masks are built like the ``sdpa`` ones: a bool keep-mask, or None when only causality
applies).  Older transformers have no registry: ``hip_attention`` is then a no-op and the model's own attention
runs (plumbing, not parity-critical — the result is the same softmax(QK^T)V in fp32).
"""
import contextlib

import torch

from . import hip

NAME = "emcid_hip"
_registered = None


def _attention_forward(module, query, key, value, attention_mask, scaling=None, dropout=0.0, is_causal=None, **kwargs):
    if dropout:
        raise hip.EmcidHipError("emcid_hip attention is inference-only (dropout must be 0)")
    if query.dtype != torch.float32:
        raise hip.EmcidHipError(f"emcid_hip attention is fp32 (the reference runs the encoder in fp32); got {query.dtype}")
    if is_causal is None:
        is_causal = getattr(module, "is_causal", False)
    causal = bool(is_causal) and query.shape[2] > 1
    return hip.attention(query, key, value, attention_mask, causal=causal, scale=scaling), None


def register() -> bool:
    global _registered
    if _registered is None:
        try:
            from transformers import AttentionInterface
            from transformers.masking_utils import AttentionMaskInterface, sdpa_mask
            AttentionInterface.register(NAME, _attention_forward)
            AttentionMaskInterface.register(NAME, sdpa_mask)
            _registered = True
        except Exception:
            _registered = False
    return _registered


class _HipQuickGELU(torch.nn.Module):
    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32:
            return hip.quick_gelu(x)
        return x * torch.sigmoid(1.702 * x)


def _swap_quick_gelu(text_encoder):
    """Replace every QuickGELU activation module (3 elementwise passes in HF) by the fused kernel; returns the
    (module, attribute, original) list to undo it."""
    undo = []
    for mod in text_encoder.modules():
        act = getattr(mod, "activation_fn", None)
        if isinstance(act, torch.nn.Module) and type(act).__name__ == "QuickGELUActivation":
            undo.append((mod, "activation_fn", act))
            mod.activation_fn = _HipQuickGELU()
    return undo


def _own_linear_forward(lin: torch.nn.Linear):
    """``lin.forward`` on the library's GEMM (clip_forward.linear: the exact-f32 MFMA kernel for fp32 HBM operands with
    K % 16 == 0, else torch — counted in LAST_PATHS either way)."""
    from . import clip_forward

    def forward(x):
        if x.is_cuda and x.dtype == torch.float32 and lin.weight.dtype == torch.float32:
            x2 = x.reshape(-1, x.shape[-1])
            if not x2.is_contiguous():
                x2 = x2.contiguous()
            return clip_forward.linear(x2, lin.weight, lin.bias).reshape(*x.shape[:-1], lin.out_features)
        return torch.nn.functional.linear(x, lin.weight, lin.bias)
    return forward


def _swap_linears(text_encoder):
    """Instance-level ``forward`` of every nn.Linear of the encoder -> the library's GEMM; returns the modules to undo."""
    from . import clip_forward
    undo = []
    if not clip_forward.OWN_GEMM:
        return undo
    for mod in text_encoder.modules():
        if type(mod) is torch.nn.Linear and "forward" not in mod.__dict__:
            mod.forward = _own_linear_forward(mod)
            undo.append(mod)
    return undo


@contextlib.contextmanager
def hip_attention(text_encoder, enabled=True, linears=True):
    """Within the block the encoder's attention and quick_gelu run on the hand-written HIP kernels, and (``linears``) its
    nn.Linear projections on the library's GEMM instead of torch's F.linear: the hooked HF forward — what an encoder the trie
    forward does not take, ``num_edit_tokens > 1`` and the cross-attention sibling use — leaves the library's kernels nowhere."""
    cfg = getattr(text_encoder, "config", None)
    on_gpu = next((p.is_cuda for p in text_encoder.parameters()), False)
    swapped = _swap_linears(text_encoder) if (enabled and linears and on_gpu) else []
    if not (enabled and cfg is not None and hasattr(cfg, "_attn_implementation") and register()):
        try:
            yield False
        finally:
            for mod in swapped:
                del mod.forward
        return
    prev = cfg._attn_implementation
    cfg._attn_implementation = NAME
    undo = _swap_quick_gelu(text_encoder)
    try:
        yield True
    finally:
        cfg._attn_implementation = prev
        for mod, name, orig in undo:
            setattr(mod, name, orig)
        for mod in swapped:
            del mod.forward
