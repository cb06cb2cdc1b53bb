#!/usr/bin/env python3
"""Does the HF tokenizers backend release the GIL inside encode_batch_fast?  Main thread counts Python loop iterations while
a helper thread encodes; compare with the count while the helper sleeps."""
import sys, time, threading
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import synthetic as syn
from emcid_amd.compute_z import expand_request_prompts, tokenize_lists
tok = syn.build_tokenizer(*syn.synthetic_vocab(syllables=True))
prompts, _, _ = expand_request_prompts(syn.make_requests(1000, names="syllable"))
tokenize_lists(tok, prompts)
bt = tok._tokenizer
def count(dur):
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < dur:
        n += 1
    return n
base = count(0.05)
res = {}
def work():
    t0 = time.perf_counter()
    for _ in range(8):
        bt.encode_batch_fast(prompts, add_special_tokens=True)
    res["encode_s"] = time.perf_counter() - t0
th = threading.Thread(target=work); th.start()
busy = count(0.05); th.join()
print({"iters_alone": base, "iters_while_encoding": busy, "ratio": round(busy / base, 2), "encode_8x_s": round(res["encode_s"], 4)})
