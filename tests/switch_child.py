"""Child process of tests/test_e2e_gpu.py::test_process_switches_leave_the_edit_unchanged: one 40-concept SD-v1.4-dims
apply_emcid_to_text_encoder under whatever EMCID_* switches the parent put into the environment (several of them are read once
per process, in C++ statics or at import); writes the edited fc2 weights' delta to argv[2] and prints the paths taken."""
import json, sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import emcid_amd
from emcid_amd import emcid_main as em, synthetic as syn
from emcid_amd.emcid_hparams import EMCIDHyperParams
from emcid_amd.nethook import get_parameter

work, out = Path(sys.argv[1]), sys.argv[2]
reqs = syn.make_requests(40, ragged=True)
hp_d = syn.sd_hparams_dict(prefix="text_model.")
names = [hp_d["rewrite_module_tmp"].format(l) for l in hp_d["layers"]]
pipe = syn.build_pipe("sd-v1.4", "cuda:0")
w0 = {n: get_parameter(pipe.text_encoder, n + ".weight").detach().clone() for n in names}
import os
for call in range(int(os.environ.get("SWITCH_CHILD_CALLS", "2"))):          # the second call runs on cached factors, graphs and planes
    for n in names:
        get_parameter(pipe.text_encoder, n + ".weight").data.copy_(w0[n])
    emcid_amd.invalidate_weight_caches(pipe.text_encoder)
    em.apply_emcid_to_text_encoder(pipe, reqs, EMCIDHyperParams(**hp_d), "cuda:0", mom2_weight=4000, edit_weight=0.5,
                                   cache_name=str(work / "cache") + "/", stats_dir=str(work / "stats"), verbose=False)
torch.cuda.synchronize()
np.savez(out, **{f"dw{i}": (get_parameter(pipe.text_encoder, n + ".weight") - w0[n]).cpu().numpy() for i, n in enumerate(names)})
print(json.dumps({k: (v if isinstance(v, (int, float, str, bool, type(None))) else str(v)) for k, v in emcid_amd.LAST_PATHS.items()}))
