#!/usr/bin/env python3
"""Host-side cost of the calls one bench step makes (are the launches keeping ahead of the GPU?).

Wraps the C-ABI wrappers the engine calls with perf_counter brackets (no device sync inside a step) and prints the
host milliseconds per step spent inside each, next to the step's wall clock."""
import sys, time, tempfile, os
from collections import defaultdict
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
from emcid_amd import emcid_main as em, hip, edit_engine, clip_forward
from emcid_amd.emcid_hparams import EMCIDHyperParams
from emcid_amd.nethook import get_parameter

acc = defaultdict(float)


def wrap(mod, name):
    f = getattr(mod, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        acc[name] += time.perf_counter() - t0
        return r
    setattr(mod, name, g)


for nm in ("factor_cov", "edit_layer_dual_apply", "gather_mean", "tree_attention", "quick_gelu"):
    wrap(hip, nm)
wrap(clip_forward, "run_layers")

device = "cuda:0"
workdir = Path(tempfile.gettempdir()) / f"emcid_bench_{os.getuid()}"
workdir.mkdir(exist_ok=True)
bench.build_inputs(1000, "cpu", workdir)
pipe, reqs, hp_d, cache, stats, layer_names = bench.build_inputs(1000, device, workdir)
hp = EMCIDHyperParams(**hp_d)
plan = em.prepare_text_encoder_edit(pipe.text_encoder, pipe.tokenizer, reqs, hp, hp.layers, hp.mom2_update_weight, stats, cache,
                                    "", verbose=False, shard=edit_engine.ConceptShard(0, 1, None))
originals = {l: get_parameter(pipe.text_encoder, plan.weight_name(l)).detach().clone() for l in bench.LAYERS}


def step():
    with torch.no_grad():
        for l in bench.LAYERS:
            get_parameter(pipe.text_encoder, plan.weight_name(l)).copy_(originals[l])
    return edit_engine.run_encoder_edit(plan, keep_factors=False, restore=False)


for _ in range(3):
    step()
torch.cuda.synchronize()
# GPU time of each call class via events on the current stream (graph mode stays on)
ev = defaultdict(list)


def wrap_ev(mod, name):
    f = getattr(mod, name)

    def g(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = f(*a, **k)
        e1.record()
        ev[name].append((e0, e1))
        return r
    setattr(mod, name, g)


orig_apply, orig_fac = hip.edit_layer_dual_apply, hip.factor_cov
wrap_ev(hip, "edit_layer_dual_apply")
wrap_ev(hip, "factor_cov")
for _ in range(5):
    step()
torch.cuda.synchronize()
for k, v in ev.items():
    print(f"GPU ms per call {k}: {sum(a.elapsed_time(b) for a, b in v) / len(v):.3f}  ({len(v) // 5} calls/step)")
hip.edit_layer_dual_apply, hip.factor_cov = orig_apply, orig_fac
if os.environ.get("SERIAL"):
    plan.side_stream = torch.cuda.current_stream()
    print("factor_cov serialized on the main stream")
for _ in range(2):
    step()
torch.cuda.synchronize()
acc.clear()
K = 10
t0 = time.perf_counter()
host = 0.0
for _ in range(K):
    h0 = time.perf_counter()
    step()
    host += time.perf_counter() - h0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"wall {wall / K * 1e3:.2f} ms/step; host time inside step() {host / K * 1e3:.2f} ms/step")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:24s} {v / K * 1e3:8.3f} ms/step host")
