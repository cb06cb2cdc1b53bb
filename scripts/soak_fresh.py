#!/usr/bin/env python3
"""Soak over DISTINCT request sets (bench.py's workload: every call a set with its own names and v* files, 26 sets cycled): per-set
weights identical on every revisit, no allocator growth, per-call wall-clock distribution.  usage: soak_fresh.py [calls=260]"""
import hashlib, os, statistics, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", str(Path(__file__).resolve().parents[1])))
os.environ.setdefault("EMCID_MANAGE_THREADS", "1")
import torch
import bench
from emcid_amd import emcid_main as em
from emcid_amd.emcid_hparams import EMCIDHyperParams
from emcid_amd.nethook import get_parameter

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 260
dev = "cuda:0"
work = Path(tempfile.gettempdir()) / f"emcid_bench_{os.getuid()}"
work.mkdir(exist_ok=True)
pipe, reqs, hp_d, cache, stats, names = bench.build_inputs(1000, dev, work)
sets = [(reqs, cache)] + [bench.request_set(1000, work, j) for j in range(1, 26)]
hp = EMCIDHyperParams(**hp_d)
orig = {n: get_parameter(pipe.text_encoder, n + ".weight").detach().clone() for n in names}
seen, walls = {}, []
mem0 = None
for i in range(n_calls):
    r, c = sets[i % len(sets)]
    with torch.no_grad():
        for n in names:
            get_parameter(pipe.text_encoder, n + ".weight").copy_(orig[n])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    em.apply_emcid_to_text_encoder(pipe, r, hp, dev, cache_name=c, stats_dir=stats, verbose=False)
    torch.cuda.synchronize()
    walls.append((time.perf_counter() - t0) * 1e3)
    h = hashlib.sha1(b"".join(get_parameter(pipe.text_encoder, n + ".weight").cpu().numpy().tobytes() for n in names)).hexdigest()
    # (the very first call of a process factors lam * C' itself and solves its first layer by block substitution, later calls
    #  multiply by the cached inverse factors: equally valid last bits, so set 0 is compared from its second visit on)
    if i > 0 and seen.setdefault(i % len(sets), h) != h:
        print("MISMATCH at call", i, "set", i % len(sets))
        sys.exit(1)
    if i == len(sets):
        mem0 = (torch.cuda.memory_allocated(), torch.cuda.memory_reserved())
from emcid_amd import edit_engine
print("host phases, ms per call:", {k: round(v / n_calls * 1e3, 3) for k, v in edit_engine.TIMING.items()})
w = sorted(walls[len(sets):])
print(f"calls {n_calls} over {len(sets)} request sets: median {statistics.median(w):.2f} p95 {w[int(0.95 * len(w))]:.2f} max {w[-1]:.2f} ms")
print("every revisit of a set gave the same weights bit for bit:", True, "| distinct results", len(set(seen.values())),
      "| alloc delta MB", (torch.cuda.memory_allocated() - mem0[0]) / 1e6, "reserved delta MB", (torch.cuda.memory_reserved() - mem0[1]) / 1e6)
