#!/usr/bin/env python3
"""Time the HF CLIP encoder forward pieces at the edit batch shape under different attention implementations."""
import sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import synthetic as syn
from emcid_amd.compute_z import build_prompt_batch
dev = "cuda:0"
pipe = syn.build_pipe("sd-v1.4", dev)
te = pipe.text_encoder
reqs = syn.make_requests(1000)
b = build_prompt_batch(pipe.tokenizer, reqs, dev)
print("batch", b.inputs["input_ids"].shape, "mask sum", int(b.inputs["attention_mask"].sum()))
def run(n=3):
    with torch.no_grad():
        for _ in range(2): te(**b.inputs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): te(**b.inputs)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for impl in ["sdpa", "eager"]:
    te.config._attn_implementation = impl
    print(impl, "full fwd ms", run())
# no attention_mask (all prompts same length?) 
inp = {"input_ids": b.inputs["input_ids"]}
te.config._attn_implementation = "sdpa"
with torch.no_grad():
    for _ in range(2): te(**inp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): te(**inp)
    torch.cuda.synchronize(); print("sdpa no-mask full fwd ms", (time.perf_counter() - t0) / 3 * 1e3)
from torch.profiler import profile, ProfilerActivity
te.config._attn_implementation = "sdpa"
with torch.no_grad(), profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    te(**b.inputs)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=70))
