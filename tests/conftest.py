import json
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

os.environ.setdefault("EMCID_MANAGE_THREADS", "1")      # the test box is a container with a CPU quota (emcid_amd.manage_threads)
REPO = Path(__file__).resolve().parents[1]
GOLDEN = REPO / "tests" / "golden"
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests must never be silently skipped on a GPU box; on CPU they are deselected by the marker.
    pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


_PRISTINE_PIPES = {}


@pytest.fixture(scope="session", autouse=True)
def _shared_pipe_builds():
    """``synthetic.build_pipe`` is a pure function of its arguments (seeded random init), and the real-dimension encoders take
    seconds to initialise (SDXL's pair: 0.8 G parameters): the suite builds each distinct pipe ONCE per device and hands every
    caller its own deep copy — same weights as a fresh build, nothing shared between tests."""
    import copy
    from emcid_amd import synthetic as syn

    build = syn.build_pipe

    def encoders(pipe):
        return {k: getattr(pipe, k) for k in ("text_encoder", "text_encoder_2") if getattr(pipe, k, None) is not None}

    def cached(kind="toy", device="cpu", sdxl=False, seed=0, syllables=False, projection_dim=None, outliers=False):
        key = (kind, str(device), sdxl, seed, syllables, projection_dim, outliers)
        if key not in _PRISTINE_PIPES:
            cpu_key = (kind, "cpu") + key[2:]
            if cpu_key not in _PRISTINE_PIPES:
                _PRISTINE_PIPES[cpu_key] = encoders(build(kind, "cpu", sdxl, seed, syllables, projection_dim, outliers))
            if key != cpu_key:
                _PRISTINE_PIPES[key] = {k: copy.deepcopy(m).to(device) for k, m in _PRISTINE_PIPES[cpu_key].items()}
        # the ENCODERS are copied; the tokenizers are built anew (a deep-copied CLIPTokenizer decodes differently: its
        # end-of-word suffix handling does not survive the copy)
        vkey = ("vocab", syllables)
        if vkey not in _PRISTINE_PIPES:
            _PRISTINE_PIPES[vkey] = syn.synthetic_vocab(syllables=syllables)
        vocab, merges = _PRISTINE_PIPES[vkey]
        toks = {"tokenizer": syn.build_tokenizer(dict(vocab), list(merges))}
        if sdxl:
            toks["tokenizer_2"] = syn.build_tokenizer(dict(vocab), list(merges))
        return syn.SyntheticPipe(**{k: copy.deepcopy(m) for k, m in _PRISTINE_PIPES[key].items()}, **toks)

    syn.build_pipe = cached
    yield
    syn.build_pipe = build
    _PRISTINE_PIPES.clear()


def load_golden(tag):
    z = np.load(GOLDEN / f"{tag}.npz")
    with open(GOLDEN / f"{tag}.json") as f:
        meta = json.load(f)
    return z, meta


def pipe_from_golden(z, kind="toy", prefix="w/", device="cpu", name="synthetic/clip-text"):
    """Rebuild the toy encoder from the fixture's stored state dict (not from the RNG)."""
    from emcid_amd import synthetic as syn

    te = syn.build_text_encoder(kind, name_or_path=name)
    sd = {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}
    te.load_state_dict(sd, strict=True)
    return te.to(device)


def write_cov_npz(stats_dir, layer_name, cov, n_samples, batch_tokens=3 * 1024):
    """Stats npz in the reference format whose moment() is exactly `cov` (count = 1)."""
    from emcid_amd import synthetic as syn

    f = syn.stats_file(stats_dir, layer_name, n_samples, batch_tokens=batch_tokens)
    f.parent.mkdir(parents=True, exist_ok=True)
    np.savez(f, **{"mom2.constructor": "util.runningstats.SecondMoment()", "mom2.count": 1,
                   "mom2.mom2": np.asarray(cov, dtype=np.float32), "sample_size": n_samples})
    return f


def write_cov_npz_model(stats_dir, layer_name, cov, n_samples, model_name):
    from emcid_amd import synthetic as syn

    f = syn.stats_file(stats_dir, layer_name, n_samples, model_name=model_name)
    f.parent.mkdir(parents=True, exist_ok=True)
    np.savez(f, **{"mom2.constructor": "util.runningstats.SecondMoment()", "mom2.count": 1,
                   "mom2.mom2": np.asarray(cov, dtype=np.float32), "sample_size": n_samples})
    return f


def xattn_from_golden(z, meta, tmp_path, device="cpu"):
    """Pipe (toy text encoder from the stored state dict + SyntheticUNet whose projections are the fixture's w_orig),
    v* cache and statistics files of the toy cross-attention fixture.  Returns (pipe, cache_name, stats_dir)."""
    from emcid_amd import synthetic as syn

    te = pipe_from_golden(z, meta["kind"], prefix="te/")
    pipe = syn.add_unet(syn.SyntheticPipe(text_encoder=te, tokenizer=syn.build_tokenizer()), meta["kind"],
                        seed=meta["unet_seed"])
    mods = dict(pipe.unet.named_modules())
    cache = str(tmp_path / "xcache") + "/"
    for li, n in enumerate(meta["layer_names"]):
        mods[n].weight.data.copy_(torch.from_numpy(z[f"w_orig/{li}"]))
        write_cov_npz_model(tmp_path / "xstats", n, z[f"cov/{li}"], meta["hparams"]["mom2_n_samples"], "unet")
    for i, r in enumerate(meta["requests"]):
        p_ = syn.xattn_vstar_cache_path(cache, r)
        p_.parent.mkdir(parents=True, exist_ok=True)
        np.savez(p_, **{n: {"v_star": z[f"vstar/{li}"][i]} for li, n in enumerate(meta["layer_names"])})
    return pipe.to(device), cache, str(tmp_path / "xstats")


def uce_pipe_from_golden(z, device="cpu"):
    """Toy text encoder + SyntheticUNet of the UCE fixture, both from its stored state dicts."""
    from emcid_amd import synthetic as syn

    te = pipe_from_golden(z, "toy", prefix="te/")
    pipe = syn.add_unet(syn.SyntheticPipe(text_encoder=te, tokenizer=syn.build_tokenizer()), "toy")
    pipe.unet.load_state_dict({k[len("unet/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("unet/")},
                              strict=True)
    return pipe.to(device)


def write_vstars(cache_name, requests, vstar, suffix=""):
    from emcid_amd import synthetic as syn

    for v, r in zip(vstar, requests):
        p = syn.vstar_cache_path(cache_name, r, suffix)
        p.parent.mkdir(parents=True, exist_ok=True)
        np.savez(p, v_star=np.asarray(v, dtype=np.float32))
