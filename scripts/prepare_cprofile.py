#!/usr/bin/env python3
"""cProfile of prepare_text_encoder_edit alone (host work up to and including the launch of the leading layers), warm.
usage: prepare_cprofile.py [n=40] [N=1000]"""
import cProfile, io, os, pstats, sys, tempfile, time, statistics
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from emcid_amd import emcid_main as em
from emcid_amd.emcid_hparams import EMCIDHyperParams

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = "cuda:0"
os.environ.setdefault("EMCID_TUNE_GEMM", "1")
work = Path(tempfile.gettempdir()) / f"emcid_bench_{os.getuid()}"
work.mkdir(exist_ok=True)
pipe, reqs, hp_d, cache, stats, names = bench.build_inputs(N, dev, work)
hp = EMCIDHyperParams(**hp_d)
em.apply_emcid_to_text_encoder(pipe, reqs, hp, dev, cache_name=cache, stats_dir=stats, verbose=False)


def prep():
    return em.prepare_text_encoder_edit(pipe.text_encoder, pipe.tokenizer, reqs, hp, hp.layers, hp.mom2_update_weight, stats, cache,
                                        "", verbose=False)


ts = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prep()
    ts.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
print("prepare returns after ms: median", round(statistics.median(ts), 3), "min", round(min(ts), 3))
pr = cProfile.Profile()
for _ in range(n):
    torch.cuda.synchronize()
    pr.enable()
    prep()
    pr.disable()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(30)
    print(s.getvalue())
