#!/usr/bin/env python3
"""Unprofiled timeline of one bench step from HIP events on the main stream: when does each edited layer's
K/Zc gather start (forward reached the layer), when does its solve start (after the wait on the factorization) and end."""
import sys, time, tempfile, os
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
from emcid_amd import emcid_main as em, hip, edit_engine
from emcid_amd.emcid_hparams import EMCIDHyperParams
from emcid_amd.nethook import get_parameter

device = "cuda:0"
workdir = Path(tempfile.gettempdir()) / f"emcid_bench_{os.getuid()}"
workdir.mkdir(exist_ok=True)
bench.build_inputs(1000, "cpu", workdir)
pipe, reqs, hp_d, cache, stats, layer_names = bench.build_inputs(1000, device, workdir)
hp = EMCIDHyperParams(**hp_d)
plan = em.prepare_text_encoder_edit(pipe.text_encoder, pipe.tokenizer, reqs, hp, hp.layers, hp.mom2_update_weight, stats, cache,
                                    "", verbose=False, shard=edit_engine.ConceptShard(0, 1, None))
originals = {l: get_parameter(pipe.text_encoder, plan.weight_name(l)).detach().clone() for l in bench.LAYERS}
marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, e))


def step():
    with torch.no_grad():
        for l in bench.LAYERS:
            get_parameter(pipe.text_encoder, plan.weight_name(l)).copy_(originals[l])
    mark("step start")
    r = edit_engine.run_encoder_edit(plan, keep_factors=False, restore=False)
    mark("step end")
    return r


for _ in range(3):
    step()
torch.cuda.synchronize()
o_gather, o_apply, o_fac, o_inv = hip.gather_mean, hip.edit_layer_dual_apply, hip.factor_cov, hip.cov_inverse
state = {"n": 0}


def gather(*a, **k):
    if state["n"] % 2 == 0:
        mark("forward reached edited layer")
    state["n"] += 1
    return o_gather(*a, **k)


def apply(*a, **k):
    mark("  solve start (after wait)")
    r = o_apply(*a, **k)
    mark("  solve end")
    return r


side_marks = []


def smark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    side_marks.append((name, e))


def fac(*a, **k):
    smark("side: factor start")
    r = o_fac(*a, **k)
    smark("side: factor end")
    return r


def inv(*a, **k):
    r = o_inv(*a, **k)
    smark("side: inverse end")
    return r


hip.gather_mean, hip.edit_layer_dual_apply, hip.factor_cov, hip.cov_inverse = gather, apply, fac, inv
marks.clear()
step()
step()
torch.cuda.synchronize()
idx = [i for i, (n, _) in enumerate(marks) if n == "step start"][-1]
t0 = marks[idx][1]
for n, e in marks[idx:]:
    print(f"{t0.elapsed_time(e):8.3f} ms  {n}")
for n, e in side_marks[-3:]:
    print(f"{t0.elapsed_time(e):8.3f} ms  {n}")

# steady state: start-to-start distance of consecutive steps (same marks, no extra instrumentation)
hip.gather_mean, hip.edit_layer_dual_apply, hip.factor_cov, hip.cov_inverse = o_gather, o_apply, o_fac, o_inv
marks.clear()
torch.cuda.synchronize()
for _ in range(8):
    step()
torch.cuda.synchronize()
starts = [e for n, e in marks if n == "step start"]
ends = [e for n, e in marks if n == "step end"]
print("start-to-start ms:", " ".join(f"{a.elapsed_time(b):.2f}" for a, b in zip(starts[:-1], starts[1:])))
print("start-to-end   ms:", " ".join(f"{a.elapsed_time(b):.2f}" for a, b in zip(starts, ends)))
