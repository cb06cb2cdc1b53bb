#!/usr/bin/env python3
"""Where the wall-clock of one apply_emcid_to_text_encoder call goes (bench.py's workload): host phases from
edit_engine.TIMING (tokenizer + subject search, v* reads, trie, launches, final sync), warm calls, median of n."""
import json, os, statistics, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from emcid_amd import emcid_main as em, edit_engine
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

def call():
    with torch.no_grad():
        for n in names:
            get_parameter(pipe.text_encoder, n + ".weight").copy_(orig[n])
    em.apply_emcid_to_text_encoder(pipe, reqs, hp, dev, cache_name=cache, stats_dir=stats, verbose=False)

t0 = time.perf_counter(); call(); torch.cuda.synchronize(); first = time.perf_counter() - t0
call(); call()
configs = ["1"]      # (round 2 compared prompt lists cut into 1 / 2 / 4 slices here; one slice won and the switch is gone)
walls = {c: [] for c in configs}
phases = {c: {} for c in configs}
for rnd in range(n_calls):              # configurations interleaved call by call: box and clock drift hit all of them alike
    for c in configs:
        edit_engine.TIMING.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        call()
        torch.cuda.synchronize()
        walls[c].append((time.perf_counter() - t0) * 1e3)
        for k, v in edit_engine.TIMING.items():
            phases[c].setdefault(k, []).append(v * 1e3)
for c in configs:
    print(json.dumps({"prep_chunks": c, "first_call_ms": round(first * 1e3, 1), "wall_ms_median": round(statistics.median(walls[c]), 2),
                      "wall_ms_min": round(min(walls[c]), 2),
                      "phases_ms_median": {k: round(statistics.median(v), 3) for k, v in phases[c].items()},
                      "cpu": bench.cpu_model(), "concepts": N}))
