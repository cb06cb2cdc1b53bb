#!/usr/bin/env python3
"""The host milliseconds before the first GPU launch of a call on a never-seen request set (bench.py's workload): the pieces of
compute_z.templated_prompt_chunk and clip_forward.build_trie, timed one by one over distinct request sets (no GPU needed for the
tokenizer half)."""
import os, statistics, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
os.environ.setdefault("EMCID_MANAGE_THREADS", "1")
import numpy as np
import torch
import bench
from emcid_amd import clip_forward, compute_z as cz, host_text, manage_threads, synthetic as syn
from emcid_amd.causal_trace import TokenRangeFinder

manage_threads()
dev = "cuda:0" if torch.cuda.is_available() else "cpu"
pipe = syn.build_pipe("sd-v1.4", "cpu", syllables=True)
tok = pipe.tokenizer
sets = [bench.request_set(1000, "/tmp/unused", i, write=False)[0] for i in range(24)]
twin = host_text.NativeClipBpe.for_tokenizer(tok)
finder = cz.finder_for(tok)
T = {}


def lap(name, t0):
    T.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
    return time.perf_counter()


for reqs in sets:
    t = time.perf_counter()
    names = list(map(cz._GET_SOURCE, reqs))
    keys = list(map(tuple, map(cz._GET_PROMPTS, reqs)))
    distinct = dict.fromkeys(keys)
    ok = set(map(type, names)) == {str}
    t = lap("request walk (names, template tuples)", t)
    key = next(iter(distinct))
    pre, suf = zip(*[p.split("{}") for p in key])
    n = len(names)
    tmpl_idx = np.tile(np.arange(len(key), dtype=np.int32), n)
    name_idx = np.repeat(np.arange(n, dtype=np.int32), len(key))
    t = lap("index arrays", t)
    packed = host_text.pack_strings(names)
    t = lap("pack_strings(names)", t)
    ids, lengths, fb = twin.encode_templated(list(pre), list(suf), packed, tmpl_idx, name_idx)
    t = lap("encode_templated (native)", t)
    S = int(lengths.max())
    ids = ids[:, :S]
    lk = finder.last_tokens(ids, names, name_idx, packed=packed)
    t = lap("last_tokens (native walk)", t)
    chunk_ids = np.ascontiguousarray(ids[:, :int(lk.max()) + 1])
    t = lap("truncate + contiguous", t)
    whole = next(cz.iter_prompt_chunks(tok, reqs, 1, defer_probe=True))
    t = lap("iter_prompt_chunks as a whole (deferred probe)", t)
    whole.verify()
    t = lap("the deferred probe itself", t)
    trie = clip_forward.build_trie(whole.ids, whole.lookup, dev, tail=np.cumsum([0] + list(whole.counts)).astype(np.int64))
    t = lap("build_trie (native + one upload)", t)
for k, v in T.items():
    print(f"{k:52s} median {statistics.median(v[4:]):7.3f} ms   min {min(v[4:]):7.3f}")
