#!/usr/bin/env python3
"""Host-side micro timings of the pieces of prepare (no GPU needed): tokenizer call variants, decode, subject walk, trie."""
import json, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from emcid_amd import synthetic as syn, clip_forward
from emcid_amd.compute_z import expand_request_prompts, tokenize_lists
from emcid_amd.causal_trace import TokenRangeFinder

tok = syn.build_tokenizer(*syn.synthetic_vocab(syllables=True))
reqs = syn.make_requests(1000, names="syllable")
def T(f, n=7):
    best = 1e9
    for _ in range(n):
        t = time.perf_counter(); r = f(); best = min(best, time.perf_counter() - t)
    return r, round(best * 1e3, 3)
out = {"TOKENIZERS_PARALLELISM": os.environ.get("TOKENIZERS_PARALLELISM")}
(prompts, subjects, counts), out["expand"] = T(lambda: expand_request_prompts(reqs))
pub, out["public_tokenizer"] = T(lambda: tok(prompts, padding=True, truncation=True))
bt = tok._tokenizer
tok(prompts[:1], padding=True, truncation=True)
encs, out["encode_batch_fast"] = T(lambda: bt.encode_batch_fast(prompts, add_special_tokens=True))
_, out["encode_batch"] = T(lambda: bt.encode_batch(prompts, add_special_tokens=True))
ids_l, out["ids_lists"] = T(lambda: [e.ids for e in encs])
_, out["np_array_ids"] = T(lambda: np.array(ids_l, dtype=np.int64))
_, out["mask_lists+array"] = T(lambda: np.array([e.attention_mask for e in encs], dtype=np.int64))
enc, out["tokenize_lists"] = T(lambda: tokenize_lists(tok, prompts))
ids = enc["input_ids"]
f = TokenRangeFinder(tok)
f.batch(ids, subjects)
lk, out["finder.batch"] = T(lambda: f.batch(ids, subjects))
rows = ids.tolist()
dec = f._decode_whole()
_, out["decode_rows"] = T(lambda: [dec(r) for r in rows])
_, out["decode_batch_backend"] = T(lambda: bt.decode_batch(rows, skip_special_tokens=False))
_, out["ids.tolist"] = T(lambda: ids.tolist())
_, out["piece_lengths+cumsum"] = T(lambda: np.cumsum(f._piece_lengths(ids), axis=1))
lookup = [r[-1] - 1 for r in lk]
_, out["build_trie"] = T(lambda: clip_forward.build_trie(ids, lookup, "cpu"))
print(json.dumps(out))
