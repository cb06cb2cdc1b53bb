#!/usr/bin/env python3
"""A few warm 1 000-concept edits (bench.py's workload, library-default GEMM selection) — target for `rocprofv3 --pmc` passes
(scripts/pmc_passes.sh): every kernel of the edit path, among them chol_step_leaf_kernel (leaf + trailing tiles + shadow
product) and the stream-K GEMMs, with the counters averaged per launch by scripts/pmc_summary.py."""
import os, sys, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
os.environ.setdefault("EMCID_TUNE_GEMM", "0")
import torch
import bench
from emcid_amd import emcid_main as em
from emcid_amd.emcid_hparams import EMCIDHyperParams

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = "cuda:0"
work = Path(tempfile.gettempdir()) / f"emcid_bench_{os.getuid()}"
work.mkdir(exist_ok=True)
pipe, reqs, hp_d, cache, stats, names = bench.build_inputs(1000, dev, work)
hp = EMCIDHyperParams(**hp_d)
for _ in range(n):
    em.apply_emcid_to_text_encoder(pipe, reqs, hp, dev, cache_name=cache, stats_dir=stats, verbose=False)
torch.cuda.synchronize()
