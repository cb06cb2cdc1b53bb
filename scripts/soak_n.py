#!/usr/bin/env python3
"""Per-call wall-clock distribution of N-concept apply_emcid_to_text_encoder calls on never-repeating request sets (cycled over
12 sets on disk): p50 / p95 / p99 / max, calls over 1.3 x the median, the slowest calls' indices.  Environment switches under
test are read by the package itself (EMCID_EARLY_VSTAR, EMCID_TOK_THREADS, EMCID_READ_THREADS ...); `--gc off` disables Python's
cyclic collector for the run, `--gc freeze` moves everything allocated so far out of its reach first.
usage: python scripts/soak_n.py N [calls] [--gc on|off|freeze]"""
import gc, os, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch, bench
from emcid_amd import emcid_main as em
from emcid_amd.emcid_hparams import EMCIDHyperParams
from emcid_amd.nethook import get_parameter

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
calls = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 400
gc_mode = sys.argv[sys.argv.index("--gc") + 1] if "--gc" in sys.argv else "on"
os.environ.setdefault("EMCID_MANAGE_THREADS", "1")
dev = "cuda:0"
work = Path(tempfile.gettempdir()) / f"emcid_bench_{os.getuid()}"
work.mkdir(exist_ok=True)
pipe, reqs, hp_d, cache, stats, names = bench.build_inputs(n, dev, work)
sets = [(reqs, cache)] + [bench.request_set(n, work, j) for j in range(1, 12)]
hp = EMCIDHyperParams(**hp_d)
orig = {k: get_parameter(pipe.text_encoder, k + ".weight").detach().clone() for k in names}


def call(i):
    r, c = sets[i % len(sets)]
    with torch.no_grad():
        for k in names:
            get_parameter(pipe.text_encoder, k + ".weight").copy_(orig[k])
    torch.cuda.synchronize()
    t = time.perf_counter()
    em.apply_emcid_to_text_encoder(pipe, r, hp, dev, cache_name=c, stats_dir=stats, verbose=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3


for i in range(10):
    call(i)
if gc_mode == "off":
    gc.disable()
elif gc_mode == "freeze":
    gc.collect()
    gc.freeze()
g0 = [s["collections"] for s in gc.get_stats()]
ts = [call(10 + i) for i in range(calls)]
g1 = [s["collections"] for s in gc.get_stats()]
v = sorted(ts)
q = lambda p: v[min(len(v) - 1, int(round(p * (len(v) - 1))))]
slow = [i for i, t in enumerate(ts) if t > 1.3 * q(0.5)]
env = {k: os.environ[k] for k in ("EMCID_EARLY_VSTAR", "EMCID_TOK_THREADS", "EMCID_READ_THREADS", "EMCID_WEIGHT_GUARD") if k in os.environ}
print(f"N {n} calls {calls} gc {gc_mode} env {env}: p50 {q(0.5):.3f} p95 {q(0.95):.3f} p99 {q(0.99):.3f} max {v[-1]:.3f} ms | p95/p50 {q(0.95) / q(0.5):.3f} | "
      f"over 1.3 x median: {len(slow)} (at {slow[:12]}) | gc collections gen0/1/2 during the run: {[b - a for a, b in zip(g0, g1)]}")
