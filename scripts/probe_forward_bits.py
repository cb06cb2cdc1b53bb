#!/usr/bin/env python3
"""Is the prefix-trie forward bit-reproducible from process to process?  One packed-trie forward of 120 ragged prompts through all
12 layers of the SD-v1.4-dims encoder; fc2 input and output of every layer (real rows) to argv[1].npz.  Compare a few runs with
`probe_forward_bits.py --compare a.npz b.npz ...`."""
import sys, hashlib
from pathlib import Path
import numpy as np
if sys.argv[1] == "--compare":
    L = [dict(np.load(f)) for f in sys.argv[2:]]
    for k in sorted(L[0], key=lambda s: (int(s[1:]), s[0])):
        hs = [hashlib.md5(l[k].tobytes()).hexdigest() for l in L]
        if len(set(hs)) > 1:
            a = L[0][k]
            for j, l in enumerate(L[1:], 1):
                diff = np.argwhere(l[k] != a)
                if len(diff):
                    print(k, f"run {j} vs 0: {len(diff)} elements differ; first at {diff[:4].tolist()}, max abs {np.abs(l[k] - a).max():.3e}; rows touched {len(set(diff[:, 0].tolist()))}")
        else:
            print(k, "identical in", len(L), "runs")
    sys.exit(0)
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import clip_forward, layer_stats as ls, synthetic as syn
pipe = syn.build_pipe("sd-v1.4", "cuda:0")
reqs = syn.make_requests(40, ragged=True)
prompts = [p.format(r["source"]) for r in reqs for p in r["prompts"]]
names = [f"text_model.encoder.layers.{i}.mlp.fc2" for i in range(12)]
graph, index = ls._packed_plan(pipe.text_encoder, names)
ids = ls.tokenize_ragged(pipe.tokenizer, prompts, 77)
out = {}
with torch.no_grad():
    for rep in range(2):
        trie, cnt = clip_forward.build_trie_packed(ids, torch.device("cuda:0"))
        n = trie.n_nodes
        def on_fc2(i, x, o):
            out[f"x{i}"] = x[:n].detach().cpu().numpy().copy()
            out[f"o{i}"] = o[:n].detach().cpu().numpy().copy()
            return o
        clip_forward.run_layers(graph, trie, 11, on_fc2, last_rows_only=False)
torch.cuda.synchronize()
np.savez(sys.argv[1], **out)
print("rows", n)
