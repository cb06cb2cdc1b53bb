"""Path constants (mirror of the reference's util/globals.py:8-38).

The reference opens ``globals.yml`` relative to the cwd at import time and crashes when it is
absent; here the file is optional (searched in the cwd, then next to this package) and the
reference's shipped values (globals.yml:3-20) are the defaults.
"""
from pathlib import Path

import yaml

_DEFAULTS = {
    "RESULTS_DIR": "results", "DATA_DIR": "data", "STATS_DIR": "data/stats",
    "XL_STATS_DIR1": "data/stats/sdxl/text1", "XL_STATS_DIR2": "data/stats/sdxl/text2",
    "KV_DIR": "kvs", "CACHE_DIR": "cache", "HPARAMS_DIR": "hparams", "EDITING_PROMPTS_CNT": 3,
    "REMOTE_ROOT_URL": "None", "RESOLUTION": 512,
}


def _load():
    data = dict(_DEFAULTS)
    for cand in (Path("globals.yml"), Path(__file__).resolve().parent / "globals.yml"):
        if cand.exists():
            with open(cand, "r") as f:
                data.update(yaml.safe_load(f) or {})
            break
    return data


_d = _load()
RESULTS_DIR, DATA_DIR, STATS_DIR, HPARAMS_DIR, KV_DIR, CACHE_DIR, XL_STATS_DIR1, XL_STATS_DIR2 = (
    Path(_d[k]) for k in ("RESULTS_DIR", "DATA_DIR", "STATS_DIR", "HPARAMS_DIR", "KV_DIR", "CACHE_DIR",
                          "XL_STATS_DIR1", "XL_STATS_DIR2"))
REMOTE_ROOT_URL = _d["REMOTE_ROOT_URL"]
RESOLUTION = _d["RESOLUTION"]
EDITING_PROMPTS_CNT = _d["EDITING_PROMPTS_CNT"]

# module-name templates of the UNet projections the cross-attention edit rewrites (reference: util/globals.py:37-38)
UNET_EDIT_TEMPLATES = {
    "cross-k": "{}.{}.attentions.{}.transformer_blocks.0.attn2.to_k",
    "cross-v": "{}.{}.attentions.{}.transformer_blocks.0.attn2.to_v",
}
