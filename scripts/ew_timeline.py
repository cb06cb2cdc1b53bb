#!/usr/bin/env python3
"""Timeline (HIP events, ms from the call's first launch) of one apply_emcid_to_text_encoder call with an edit_weight the process
has not used: when the side stream's factorization of the four lam * C' starts and ends, when each layer's inverse factor is
there, when each layer's solve starts (after its wait) and ends; the same for a warm call beside it."""
import copy, os, sys, tempfile, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
os.environ.setdefault("EMCID_MANAGE_THREADS", "1")
import bench
from emcid_amd import emcid_main as em, hip, edit_engine, clip_forward
from emcid_amd.emcid_hparams import EMCIDHyperParams
from emcid_amd.nethook import get_parameter

dev = "cuda:0"
work = Path(tempfile.gettempdir()) / f"emcid_bench_{os.getuid()}"
work.mkdir(exist_ok=True)
pipe, reqs, hp_d, cache, stats, names = bench.build_inputs(1000, dev, work)
hp = EMCIDHyperParams(**hp_d)
orig = {n: get_parameter(pipe.text_encoder, n + ".weight").detach().clone() for n in names}
marks = []


def mark(name, stream=None):
    e = torch.cuda.Event(enable_timing=True)
    e.record(stream if stream is not None else torch.cuda.current_stream())
    marks.append((name, e))


o_fac, o_inv, o_apply, o_prefix = hip.factor_cov, hip.cov_inverse, hip.edit_layer_dual_apply, clip_forward.run_prefix


def fac(*a, **k):
    mark("side: factorization starts")
    r = o_fac(*a, **k)
    mark("side: factorization queued to here")
    return r


def inv(cf, first, count):
    r = o_inv(cf, first, count)
    mark(f"side: inverse factor of layer {first}..{first + count - 1} done")
    return r


def app(*a, **k):
    mark("main: solve starts behind its waits")
    r = o_apply(*a, **k)
    mark("main: solve ends")
    return r


def prefix(*a, **k):
    mark("main: first launch of the call")
    r = o_prefix(*a, **k)
    mark("main: leading layers done")
    return r


hip.factor_cov, hip.cov_inverse, hip.edit_layer_dual_apply, clip_forward.run_prefix = fac, inv, app, prefix


def call(ew=None):
    with torch.no_grad():
        for n in names:
            get_parameter(pipe.text_encoder, n + ".weight").copy_(orig[n])
    marks.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    em.apply_emcid_to_text_encoder(pipe, reqs, copy.deepcopy(hp), dev, cache_name=cache, stats_dir=stats, verbose=False, edit_weight=ew)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    base = next(e for n, e in marks if n.startswith("main: first launch"))
    print(f"--- edit_weight {ew}: wall {wall:.2f} ms")
    for n, e in marks:
        print(f"  {base.elapsed_time(e):8.3f} ms  {n}")


for _ in range(3):
    call()
for ew in (float(x) for x in (sys.argv[1:] or ["0.37", "0.37", "0.43"])):
    call(ew)
