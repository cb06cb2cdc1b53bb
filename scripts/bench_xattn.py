#!/usr/bin/env python3
"""Cross-attention K/V edit (reference emcid_main.py:314-548) on ONE MI355X: SD-v1.4 shapes (text encoder 768/3072/12L,
32 projections 768 -> 320/640/1280), N concepts, v* and statistics cached on disk.  Whole apply_* calls are timed
(host tokenization + cache reads included; the second call has COV_CACHE warm).  One JSON line."""
import copy, json, shutil, sys, tempfile, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import emcid_main as em, synthetic as syn
from emcid_amd.emcid_hparams import EMCIDHyperParams
from emcid_amd.layer_stats import get_all_cross_attn_kv_layer_names

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = "cuda:0"
kind = "sd-v1.4"
tmp = Path(tempfile.mkdtemp())
pipe = syn.add_unet(syn.build_pipe(kind, dev, syllables=True), kind)
names = get_all_cross_attn_kv_layer_names(pipe)
dims = {n: dict(pipe.unet.named_modules())[n].out_features for n in names}
reqs = syn.make_requests(N, names="syllable")
cache = str(tmp / "cache") + "/"
syn.write_xattn_vstar_cache(cache, reqs, dims, seed=6, scale=0.5)
syn.write_stats_cache(tmp / "stats", names[:1], 768, 100, seed=2, t=1536, model_name="unet")
first = syn.stats_file(tmp / "stats", names[0], 100, model_name="unet")
for n in names[1:]:        # every projection sees the same text embeddings: same statistics under every name
    shutil.copy(first, syn.stats_file(tmp / "stats", n, 100, model_name="unet"))
hp_d = syn.sd_hparams_dict(mom2_update_weight=4000, mom2_n_samples=100)
w0 = {k: v.clone() for k, v in pipe.unet.state_dict().items()}
times, load_s = [], []
_load = em.load_v_stars_cross_attn


def timed_load(*a, **k):
    t = time.perf_counter()
    r = _load(*a, **k)
    load_s.append(time.perf_counter() - t)
    return r


em.load_v_stars_cross_attn = timed_load
for it in range(4):
    pipe.unet.load_state_dict(w0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    em.apply_emcid_to_cross_attn(pipe, reqs, EMCIDHyperParams(**hp_d), dev, cache_name=cache, stats_dir=str(tmp / "stats"),
                                 verbose=False)
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
print(json.dumps({"workload": f"cross-attention K/V edit, {N} concepts, 32 projections, SD-v1.4 shapes",
                  "first_call_ms": times[0] * 1e3, "warm_call_ms": min(times[1:]) * 1e3,
                  "of_which_vstar_npz_reads_ms": min(load_s[1:]) * 1e3,
                  "warm_call_without_vstar_reads_ms": (min(times[1:]) - min(load_s[1:])) * 1e3,
                  "concept_edits_per_s_warm_call": N / min(times[1:])}))
