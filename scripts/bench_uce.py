#!/usr/bin/env python3
"""UCE closed-form edits (reference emcid/uce_train.py) on ONE MI355X at SD-v1.4 shapes: N (old -> new) concept pairs,
two retained texts.  `te`: fc2 of text-encoder layer 11 (3072 -> 768); `ca`: the 32 cross-attention projections
(768 -> 320/640/1280).  Whole calls are timed (tokenization, encoder forward, closed form).  One JSON line."""
import json, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import synthetic as syn, uce_train as uce

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = "cuda:0"
pipe = syn.add_unet(syn.build_pipe("sd-v1.4", dev, syllables=True), "sd-v1.4")
old = [r["source"] for r in syn.make_requests(N, names="syllable")]
new = ["a realist artist"] * N
retain = ["painting", "a photo of the artist"]
te0 = {k: v.clone() for k, v in pipe.text_encoder.state_dict().items()}
un0 = {k: v.clone() for k, v in pipe.unet.state_dict().items()}
out = {"workload": f"UCE closed form, {N} concept pairs + 2 retained texts, SD-v1.4 shapes, technique 'tensor'"}
for tag, fn in (("te", lambda: uce.edit_text_encoder_uce(pipe, old, new, retain, layer_to_edit=11)),
                ("ca", lambda: uce.edit_model_uce(pipe, old, new, retain))):
    times, runs = [], []
    for it in range(3):
        pipe.text_encoder.load_state_dict(te0)
        pipe.unet.load_state_dict(un0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        runs.append(dict(uce.LAST_RUN))
    best = min(range(1, 3), key=lambda i: times[i])
    out[tag] = {"first_call_ms": times[0] * 1e3, "warm_call_ms": times[best] * 1e3,
                "forward_and_gather_ms": runs[best]["forward_s"] * 1e3, "closed_form_ms": runs[best]["solve_s"] * 1e3,
                "context_rows": runs[best]["rows"], "edits_per_s_warm_call": N / times[best]}
print(json.dumps(out))
