#!/usr/bin/env python3
"""Per-call wall times of the headline edit (the bench's `call`), with the garbage collector's pauses attributed: prints the
sorted per-call times, the calls during which a collection ran, and the same loop with the collector frozen/disabled."""
import gc, json, os, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from emcid_amd import emcid_main as em
from emcid_amd.emcid_hparams import EMCIDHyperParams
from emcid_amd.edit_engine import ConceptShard
from emcid_amd.nethook import get_parameter

os.environ.setdefault("EMCID_TUNE_GEMM", "1")
n = int(os.environ.get("CALLS", "40"))
workdir = Path(tempfile.gettempdir()) / f"emcid_bench_{os.getuid()}"
workdir.mkdir(exist_ok=True, mode=0o700)
bench.build_inputs(1000, "cpu", workdir)
pipe, reqs, hp_d, cache, stats, names = bench.build_inputs(1000, "cuda:0", workdir)
hp = EMCIDHyperParams(**hp_d)
orig = {k: get_parameter(pipe.text_encoder, k + ".weight").detach().clone() for k in names}
shard = ConceptShard(0, 1, None)

def call():
    with torch.no_grad():
        for k in names:
            get_parameter(pipe.text_encoder, k + ".weight").copy_(orig[k])
    em.apply_emcid_to_text_encoder(pipe, reqs, hp, "cuda:0", cache_name=cache, stats_dir=stats, verbose=False, shard=shard)

events = []
def on_gc(phase, info):
    if phase == "start":
        on_gc.t = time.perf_counter()
    else:
        events.append((info["generation"], (time.perf_counter() - on_gc.t) * 1e3))
gc.callbacks.append(on_gc)

from emcid_amd import edit_engine
phases = []

def loop(tag):
    per, gcs = [], []
    torch.cuda.synchronize()
    for _ in range(n):
        k = len(events)
        edit_engine.TIMING.clear()
        t = time.perf_counter(); call(); per.append((time.perf_counter() - t) * 1e3)
        phases.append((per[-1], {k_: round(v * 1e3, 2) for k_, v in edit_engine.TIMING.items()}))
        gcs.append([(g, round(ms, 2)) for g, ms in events[k:]])
    torch.cuda.synchronize()
    slow = [(i, round(p, 1), g) for i, (p, g) in enumerate(zip(per, gcs)) if p > 1.3 * sorted(per)[len(per) // 2]]
    print(json.dumps({"mode": tag, "mean": round(sum(per) / n, 2), "median": round(sorted(per)[n // 2], 2),
                      "min": round(min(per), 2), "max": round(max(per), 2), "sorted": [round(p, 1) for p in sorted(per)],
                      "slow_calls(index, ms, gc events)": slow}))

for _ in range(4):
    call()
loop("default")
for ms, ph in phases[:9]:
    print(json.dumps({"call_ms": round(ms, 1), "phases": ph}))
if os.environ.get("GC_MODES"):
    gc.collect(); gc.freeze()
    loop("gc.freeze")
    gc.disable()
    loop("gc.disable")
