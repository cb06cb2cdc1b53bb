#!/usr/bin/env python3
"""The native templated encode alone (1 000 never-seen 3-syllable names x 3 templates, the bench's shape): ms per C call."""
import sys; from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import time, numpy as np
from emcid_amd import synthetic as syn, host_text
pipe = syn.build_pipe("toy","cpu",syllables=True)
twin = host_text.NativeClipBpe.for_tokenizer(pipe.tokenizer)
sets=[[r["source"] for r in syn.make_requests(1000,names="syllable",name_seed=3+101*i)] for i in range(40)]
pre=["painting by ","artwork by ","style of "]; suf=["","",""]
def med(f,n=400):
    ts=[]
    for i in range(n):
        t=time.perf_counter(); f(i); ts.append((time.perf_counter()-t)*1e3)
    ts.sort(); return round(ts[len(ts)//2],4), round(ts[len(ts)//10],4), round(ts[0],4)
packed=[host_text.pack_strings(s) for s in sets]
lib=twin._lib
pb,po=host_text.pack_strings(pre); sb,so=host_text.pack_strings(suf)
P=host_text._ptr
n=3000
tmpl_idx=np.tile(np.arange(3,dtype=np.int32),1000); name_idx=np.repeat(np.arange(1000,dtype=np.int32),3)
ids=np.empty((n,20),dtype=np.int64); lengths=np.empty(n,dtype=np.int32); fb=np.empty(n,dtype=np.uint8); nl=np.empty(n,dtype=np.int32)
def raw(i):
    nb,no=packed[i%40]
    lib.emcid_bpe_encode_templated(twin._h,pb,P(po),sb,P(so),3,nb,P(no),1000,P(tmpl_idx),P(name_idx),n,twin.bos,twin.eos,twin.pad,20,P(ids),P(lengths),P(fb),P(nl))
print("emcid_bpe_encode_templated: median, p10, min ms", med(raw))
