"""emcid_amd — MI355X-native implementation of EMCID's closed-form mass-edit hot path.

Public surface mirrors the reference's (SilentView/EMCID) for this path:
    emcid_amd.emcid_main   apply_emcid_to_text_encoder, apply_emcid_to_sdxl_text_encoders, execute_*,
                           get_cov_text_encoder, apply_emcid_to_model,
                           apply_emcid_to_cross_attn, execute_emcid_cross_attn (UNet cross-attention K/V)
    emcid_amd.uce_train    edit_text_encoder_uce, edit_model_uce (the UCE baseline's closed forms)
    emcid_amd.layer_stats  layer_stats_text_encoder (Stage 0)
    emcid_amd.compute_ks / compute_z / runningstats / nethook / stat_dataset / causal_trace / emcid_hparams
The compute is in csrc/ (hand-written HIP for gfx950) behind the C ABI declared in include/emcid_hip.h.
"""
__version__ = "0.1.0"

from .emcid_hparams import EMCIDHyperParams, EMCIDXLHyperParams  # noqa: F401


def __getattr__(name):
    # lazy: importing the package must not require a built kernel library
    if name in ("apply_emcid_to_text_encoder", "apply_emcid_to_sdxl_text_encoders", "apply_emcid_to_model",
                "execute_emcid_text_encoder", "execute_emcid_sd_xl_text_encoders", "get_cov_text_encoder",
                "apply_emcid_to_cross_attn", "execute_emcid_cross_attn", "get_cov_cross_attn"):
        from . import emcid_main
        return getattr(emcid_main, name)
    if name in ("edit_text_encoder_uce", "edit_model_uce"):
        from . import uce_train
        return getattr(uce_train, name)
    raise AttributeError(name)


def _cap_tokenizer_threads():
    """The HF `tokenizers` backend (Rust, rayon) starts one worker per hardware thread; on a 128-core editing host a
    3 000-prompt batch then spends more time waking 128 workers than encoding (measured on 2 x EPYC 9575F:
    encode_batch_fast 7.4 ms with 128 threads, 4.9 ms with 32; the whole tokenizer step of a 1 000-concept edit 15.3 -> 8.8 ms).
    rayon reads RAYON_NUM_THREADS when its pool is first used, so a DEFAULT is set here, at import, if the process has
    none (EMCID_RAYON_THREADS=0 leaves the environment alone, any other value is used instead of 32)."""
    import os
    want = os.environ.get("EMCID_RAYON_THREADS", "32")
    if want != "0" and "RAYON_NUM_THREADS" not in os.environ and (os.cpu_count() or 1) > int(want):
        os.environ["RAYON_NUM_THREADS"] = want


_cap_tokenizer_threads()
