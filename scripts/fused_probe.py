import os, sys, tempfile
from pathlib import Path
import torch
sys.path.insert(0, "/root/repo")
from emcid_amd import clip_forward as cf, edit_engine as ee, emcid_main as em, synthetic as syn
from emcid_amd.emcid_hparams import EMCIDHyperParams
from emcid_amd.nethook import get_parameter
DEV = "cuda:0"
kind, layers = "sd-v1.4", (7, 8, 9, 10)
n_req = int(sys.argv[1]) if len(sys.argv) > 1 else 40
hidden, inter = syn.ENCODER_DIMS[kind][:2]
tmp = Path(tempfile.mkdtemp())
reqs = syn.make_requests(n_req, ragged=True, names="syllable")
hp_d = syn.sd_hparams_dict(layers=layers, mom2_update_weight=60, mom2_n_samples=100)
names = [hp_d["rewrite_module_tmp"].format(l) for l in layers]
cache = str(tmp / "cache") + "/"
syn.write_vstar_cache(cache, reqs, hidden, seed=1, scale=0.5)
syn.write_stats_cache(tmp / "stats", names, inter, 100, seed=2, t=2 * inter)
em.clear_caches()
pipe = syn.build_pipe(kind, DEV, syllables=True)
w0 = {n: get_parameter(pipe.text_encoder, n + ".weight").detach().clone() for n in names}
out = {}
for mode in ("cold", "fused", "stages", "fused2", "stages2"):
    os.environ["EMCID_FUSED_EDIT_LAYER"] = "0" if mode.startswith("stages") else "1"
    with torch.no_grad():
        for n in names:
            get_parameter(pipe.text_encoder, n + ".weight").copy_(w0[n])
    hp = EMCIDHyperParams(**hp_d)
    plan = em.prepare_text_encoder_edit(pipe.text_encoder, pipe.tokenizer, reqs, hp, hp.layers, 60, str(tmp / "stats"), cache, verbose=False)
    edits = ee.run_encoder_edit(plan, trace=True)
    ee.check_info(plan)
    out[mode] = edits
for a, b in (("fused", "stages"), ("fused", "fused2"), ("stages", "stages2"), ("cold", "stages")):
    line = [f"{a} vs {b}:"]
    for ea, eb in zip(out[a], out[b]):
        line.append(f"L{ea.layer} K {(ea.K - eb.K).abs().max().item():.1e} Zc {(ea.Zc - eb.Zc).abs().max().item():.1e} dW {(ea.dW - eb.dW).abs().max().item() / ea.dW.abs().max().item():.1e}")
    print(" | ".join(line))
