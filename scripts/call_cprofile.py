#!/usr/bin/env python3
"""cProfile of warm apply_emcid_to_text_encoder calls (bench.py's workload): where the host's Python time goes.
usage: call_cprofile.py [n_calls=30] [N=1000]"""
import cProfile, io, os, pstats, sys, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from emcid_amd import emcid_main as em
from emcid_amd.emcid_hparams import EMCIDHyperParams
from emcid_amd.nethook import get_parameter

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 30
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


for _ in range(3):
    call()
pr = cProfile.Profile()
pr.enable()
for _ in range(n_calls):
    call()
pr.disable()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue())
