import sys; from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import time, os, numpy as np, torch
from emcid_amd import synthetic as syn, compute_z as cz, clip_forward as cf, host_text
pipe = syn.build_pipe("toy","cpu",syllables=True)
tok = pipe.tokenizer
def bench(fn, n=200):
    ts=[]
    for _ in range(n):
        t=time.perf_counter(); r=fn(); ts.append((time.perf_counter()-t)*1e3)
    ts.sort(); return ts[len(ts)//2], ts[0], r
sets=[syn.make_requests(1000,names="syllable",name_seed=3+101*i) for i in range(60)]
i=[0]
def prep():
    reqs=sets[i[0]%60]; i[0]+=1
    it=cz.iter_prompt_chunks(tok, reqs, 1, defer_probe=True)
    return next(it)
m,mn,pc=bench(prep); print("iter_prompt_chunks median %.3f min %.3f ms"%(m,mn), "TOK_THREADS", os.environ.get("EMCID_TOK_THREADS"), pc.ids.shape)
def trie():
    return cf.build_trie(pc.ids, pc.lookup, "cpu", tail=np.cumsum([0]+list(pc.counts)).astype(np.int64))
m,mn,t=bench(trie); print("build_trie median %.3f min %.3f ms"%(m,mn), t.n_nodes)
