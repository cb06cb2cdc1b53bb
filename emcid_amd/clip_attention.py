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


@contextlib.contextmanager
def hip_attention(text_encoder, enabled=True):
    """Within the block the encoder's attention (and quick_gelu) run on the hand-written HIP kernels."""
    cfg = getattr(text_encoder, "config", None)
    if not (enabled and cfg is not None and hasattr(cfg, "_attn_implementation") and register()):
        yield False
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
