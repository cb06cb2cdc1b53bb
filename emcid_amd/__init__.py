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
    if name in ("invalidate_weight_caches", "StaleWeightCacheError"):
        # after rewriting an encoder weight through ``param.data`` (which torch's version counter does not see) call
        # ``emcid_amd.invalidate_weight_caches(text_encoder)``; without it the content guard catches dense rewrites at the next
        # edit call and that call is redone ("weights written behind the caches")
        from . import clip_forward
        return getattr(clip_forward, name)
    if name == "LAST_PATHS":
        # which forward / GEMM path this process's calls took (counters; clip_forward.LAST_PATHS): tests and bench.py assert
        # "own kernels" on it instead of inferring that from a profiler's kernel table
        from . import clip_forward
        return clip_forward.LAST_PATHS
    raise AttributeError(name)


def effective_cpu_count() -> int:
    """CPUs this process may actually use: the scheduler affinity mask AND the container's CPU bandwidth quota (cgroup v2
    ``cpu.max`` / v1 ``cpu.cfs_quota_us``), which ``os.cpu_count()`` and PyTorch's default thread count both ignore."""
    import math
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "<quota|max> <period>"
            q, period = f.read().split()[:2]
            if q != "max":
                quota = int(q) / int(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, period = int(f.read()), int(g.read())
                if q > 0 and period > 0:
                    quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, math.floor(quota)))
    return max(1, n)


def _respect_cpu_quota():
    """PyTorch sizes its intra-op (OpenMP) pool by the machine's cores; inside a container with a CPU quota (the MI355X
    boxes here: 256 hardware threads visible, quota 16) the 128 OpenMP workers spin after every small CPU tensor op of the
    edit's host side, use up the quota within the scheduler's 100 ms period, and the whole process — the thread feeding the
    GPU included — is frozen for the rest of it: every third 19 ms edit took 62 ms (scripts/call_jitter.py,
    scripts/cgroup_probe.sh: 62 s throttled in a 15 s run).  The pool is therefore LOWERED to the CPUs the process can
    actually use, never raised; EMCID_TORCH_THREADS=0 leaves PyTorch alone, any other value is used instead."""
    import os
    want = os.environ.get("EMCID_TORCH_THREADS", "")
    if want == "0" or "OMP_NUM_THREADS" in os.environ:
        return
    try:
        import torch
        n = int(want) if want else effective_cpu_count()
        if n >= 1 and torch.get_num_threads() > n:
            torch.set_num_threads(n)
    except Exception:       # never fail an import over a tuning default
        pass


def _cap_tokenizer_threads():
    """The HF `tokenizers` backend (Rust, rayon) starts one worker per hardware thread; on a 128-core editing host a
    3 000-prompt batch then spends more time waking 128 workers than encoding (measured on 2 x EPYC 9575F:
    encode_batch_fast 7.4 ms with 128 threads, 4.9 ms with 32; the whole tokenizer step of a 1 000-concept edit 15.3 -> 8.8 ms).
    rayon reads RAYON_NUM_THREADS when its pool is first used, so a DEFAULT is set here, at import, if the process has
    none (EMCID_RAYON_THREADS=0 leaves the environment alone, any other value is used instead of 32)."""
    import os
    want = os.environ.get("EMCID_RAYON_THREADS", "")
    if want == "0" or "RAYON_NUM_THREADS" in os.environ:
        return
    n = int(want) if want else min(32, effective_cpu_count())
    if (os.cpu_count() or 1) > n:
        os.environ["RAYON_NUM_THREADS"] = str(n)


_THREADS_MANAGED = False


def manage_threads(force: bool = False) -> bool:
    """Opt-in host-thread hygiene for an editing process: lower PyTorch's intra-op pool to the CPUs the container may use
    (_respect_cpu_quota) and give the `tokenizers` backend a bounded rayon pool (_cap_tokenizer_threads).  Nothing happens
    at import: a host application that embeds this package keeps its own settings.  The first ``prepare_*`` call of a
    process runs this when EMCID_MANAGE_THREADS=1 — which bench.py, the run_emcid CLI and the test suite set by default
     — or call it yourself.  Returns whether it acted."""
    global _THREADS_MANAGED
    import os
    if _THREADS_MANAGED or not (force or os.environ.get("EMCID_MANAGE_THREADS", "0") == "1"):
        return False
    _THREADS_MANAGED = True
    _cap_tokenizer_threads()
    _respect_cpu_quota()
    return True
