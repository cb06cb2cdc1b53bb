#!/usr/bin/env python3
"""bench.sdxl_record with the fused edited-layer call on / off (EMCID_FUSED_EDIT_LAYER)."""
import os, sys, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
work = Path(tempfile.gettempdir()) / f"emcid_bench_{os.getuid()}"
work.mkdir(exist_ok=True)
for v in ("1", "0", "1"):
    os.environ["EMCID_FUSED_EDIT_LAYER"] = v
    r = bench.sdxl_record(work, "cuda:0")
    print("fused", v, {k: r[k] for k in ("ms_per_call_median", "ms_per_call") if k in r}, flush=True)
