#!/usr/bin/env python3
"""Stage 0 (BASELINE.json config 5) on one MI355X: fc2-input second moment of ALL 12 layers of the SD-v1.4-dim
text encoder over a synthetic caption set, single pass (layer_stats_text_encoder_multi).  Prints one JSON line:
tokens/s over the whole job, the SYRK kernel's share and its fraction of the fp32 MFMA peak."""
import argparse, json, os, sys, tempfile, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip, synthetic as syn
from emcid_amd import layer_stats as ls
from emcid_amd.layer_stats import layer_stats_text_encoder_multi

ap = argparse.ArgumentParser()
ap.add_argument("--captions", type=int, default=20000)
ap.add_argument("--layers", type=int, default=12)
ap.add_argument("--workers", type=int, default=8)
ap.add_argument("--kind", default="sd-v1.4")
a = ap.parse_args()
tmp = Path(tempfile.mkdtemp())
caps = syn.write_captions(tmp / "data" / "ccs_filtered.json", a.captions, seed=2)
pipe = syn.build_pipe(a.kind, "cuda:0")
names = [f"text_model.encoder.layers.{i}.mlp.fc2" for i in range(a.layers)]
d = syn.ENCODER_DIMS[a.kind][1]
# warm-up on a small sample (kernel load, allocator)
layer_stats_text_encoder_multi(pipe.text_encoder, pipe.tokenizer, names[:2], tmp / "warm", sample_size=500,
                               data_path=str(tmp / "data" / "ccs_filtered.json"), progress=None, num_workers=0)
torch.cuda.synchronize()
hip.profile_enable(["gram"])
t0 = time.perf_counter()
stats = layer_stats_text_encoder_multi(pipe.text_encoder, pipe.tokenizer, names, tmp / "stats", sample_size=a.captions,
                                       data_path=str(tmp / "data" / "ccs_filtered.json"), progress=None,
                                       num_workers=a.workers)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
prof = hip.profile_collect()
hip.profile_enable([])
tokens = stats[names[0]].mom2.count
gram_ms, gram_launches = prof.get("gram", (0.0, 0))
syrk_flops = float(tokens) * d * d * a.layers          # T d^2 per layer (algorithmic: what the job is worth)
rows = ls.LAST_RUN.get("rows", tokens)                  # distinct prefixes actually pushed through the kernel (packed forward)
exec_flops = float(rows) * d * d * a.layers
print(json.dumps({
    "stage": 0, "captions": a.captions, "layers": a.layers, "tokens": tokens, "wall_s": wall,
    "tokens_per_s": tokens / wall, "layer_tokens_per_s": tokens * a.layers / wall,
    "gram_ms": gram_ms, "gram_launches": gram_launches, "gram_share_of_wall": gram_ms * 1e-3 / wall,
    "forward": ls.LAST_RUN.get("forward", "hooked-hf"), "gram_rows": rows,
    "gram_tflops_algorithmic": syrk_flops / (gram_ms * 1e-3) / 1e12 if gram_ms else None,
    "gram_tflops_executed": exec_flops / (gram_ms * 1e-3) / 1e12 if gram_ms else None,
    "gram_frac_f32_mfma_peak": exec_flops / (gram_ms * 1e-3) / 157.3e12 if gram_ms else None,
    "workers": a.workers}))
