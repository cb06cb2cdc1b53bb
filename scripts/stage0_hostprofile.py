#!/usr/bin/env python3
"""Where the wall-clock of Stage 0 (100 000 captions, 12 layers, one pass) goes on the host: cProfile of
layer_stats_text_encoder_multi plus the device-busy time of the same run (all kernel classes bracketed by events)."""
import cProfile, io, pstats, sys, tempfile, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip, synthetic as syn
from emcid_amd.layer_stats import layer_stats_text_encoder_multi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
tmp = Path(tempfile.mkdtemp())
syn.write_captions(tmp / "data" / "ccs_filtered.json", n, seed=2)
pipe = syn.build_pipe("sd-v1.4", "cuda:0")
names = [f"text_model.encoder.layers.{i}.mlp.fc2" for i in range(12)]
layer_stats_text_encoder_multi(pipe.text_encoder, pipe.tokenizer, names[:2], tmp / "warm", sample_size=500,
                               data_path=str(tmp / "data" / "ccs_filtered.json"), progress=None, num_workers=0)
torch.cuda.synchronize()
for prof_on in (False, True):
    hip.profile_enable(hip.PROF_CLASSES if prof_on else [])
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    if not prof_on:
        pr.enable()
    layer_stats_text_encoder_multi(pipe.text_encoder, pipe.tokenizer, names, tmp / f"stats{int(prof_on)}", sample_size=n,
                                   data_path=str(tmp / "data" / "ccs_filtered.json"), progress=None, num_workers=8)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if not prof_on:
        pr.disable()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
        print(f"wall {wall:.2f} s (under cProfile)")
        print("\n".join(s.getvalue().splitlines()[:40]))
    else:
        prof = hip.profile_collect()
        print(f"wall {wall:.2f} s with event brackets; device ms by class:", {k: round(v[0]) for k, v in prof.items()},
              "sum", round(sum(v[0] for v in prof.values())))
