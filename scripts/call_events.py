#!/usr/bin/env python3
"""Host and device timelines of one apply_emcid_to_text_encoder call side by side, without a profiler: HIP events are recorded
on the launch stream around the prefix forward and every layer solve, together with the host clock at the moment each was
enqueued.  An event's device time can never be earlier than its enqueue time, so `dev - host` ~ 0 means the GPU was waiting for
the host at that point (host-bound) and a large value means the host was ahead (device-bound).  Median over n warm calls.
usage: call_events.py [n_calls=10] [N=1000]"""
import json, os, statistics, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from emcid_amd import emcid_main as em, edit_engine, clip_forward, hip
from emcid_amd.emcid_hparams import EMCIDHyperParams
from emcid_amd.nethook import get_parameter

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = "cuda:0"
os.environ.setdefault("EMCID_TUNE_GEMM", "1")
work = Path(tempfile.gettempdir()) / f"emcid_bench_{os.getuid()}"
work.mkdir(exist_ok=True)
pipe, reqs, hp_d, cache, stats, names = bench.build_inputs(N, dev, work)
hp = EMCIDHyperParams(**hp_d)
orig = {n: get_parameter(pipe.text_encoder, n + ".weight").detach().clone() for n in names}

marks = []          # (label, host seconds, event)


def mark(label):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record(torch.cuda.current_stream())
    marks.append((label, time.perf_counter(), ev))


def wrap(mod, name, label):
    real = getattr(mod, name)
    count = [0]

    def f(*a, **k):
        i = count[0]
        count[0] += 1
        mark(f"{label}[{i}] enqueue begins")
        out = real(*a, **k)
        mark(f"{label}[{i}] enqueued")
        return out
    f.count = count
    setattr(mod, name, f)
    return f


w_prefix = wrap(clip_forward, "run_prefix", "prefix")
w_solve = wrap(hip, "edit_layer_dual_apply", "solve")
w_fused = wrap(hip, "clip_edit_layer_tail", "edit layer (one C call)")          # the warm single-rank path
w_head = wrap(hip, "clip_layer_head", "attention + fc1")
w_join = wrap(edit_engine.EncoderEditPlan, "resolve_targets", "v* rows")


def call():
    with torch.no_grad():
        for n in names:
            get_parameter(pipe.text_encoder, n + ".weight").copy_(orig[n])
    em.apply_emcid_to_text_encoder(pipe, reqs, hp, dev, cache_name=cache, stats_dir=stats, verbose=False)


for _ in range(3):
    call()
torch.cuda.synchronize()
table = {}
walls = []
for _ in range(n_calls):
    marks.clear()
    w_prefix.count[0] = w_solve.count[0] = w_fused.count[0] = w_head.count[0] = w_join.count[0] = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mark("call start")
    call()
    mark("call returns")
    torch.cuda.synchronize()
    walls.append((time.perf_counter() - t0) * 1e3)
    e0 = marks[0][2]
    for label, th, ev in marks:
        table.setdefault(label, []).append(((th - t0) * 1e3, e0.elapsed_time(ev)))
print(f"wall ms median {statistics.median(walls):.2f} min {min(walls):.2f}")
print(f"{'mark':34s} {'host ms':>9s} {'device ms':>10s} {'dev-host':>9s}")
for label, v in table.items():
    h = statistics.median(x[0] for x in v)
    d = statistics.median(x[1] for x in v)
    print(f"{label:34s} {h:9.2f} {d:10.2f} {d - h:9.2f}")
