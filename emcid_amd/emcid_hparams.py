"""Hyper-parameter records of the text-encoder edit path.

Schema-compatible with the JSON files the reference loads through ``HyperParams.from_json``
(reference: util/hparams.py:11-16; emcid/emcid_hparams.py:55-163 ``EMCIDHyperParams``, :166-276
``EMCIDXLHyperParams``; shipped files hparams/*.json): same keys, same defaults, mutable in place (the entry points
overwrite ``mom2_update_weight`` / ``edit_weight``, emcid_main.py:846-847).  The edit path itself reads only the
fields of ``_EditFields``; the Stage-1 knobs are carried so that every shipped file loads unchanged.

Records are keyword-only (``cls(**json)`` is the only way the reference builds them), which lets the SD and SDXL
records share one field list instead of repeating it.
"""
import json
from dataclasses import dataclass, fields
from typing import Any, List, Optional


@dataclass(kw_only=True)
class HyperParams:
    @classmethod
    def from_json(cls, fpath):
        with open(fpath, "r") as f:
            return cls(**json.load(f))

    @classmethod
    def from_dict(cls, d):
        return cls(**d)

    def to_dict(self):
        return {f.name: getattr(self, f.name) for f in fields(self)}

    def to_json(self, fpath):
        with open(fpath, "w") as f:
            json.dump(self.to_dict(), f, indent=4)


@dataclass(kw_only=True)
class _EditFields(HyperParams):
    # ---- read by the closed-form path -------------------------------------------------------------------
    layers: List[int]                 # edited encoder layers, forward order
    mom2_update_weight: int           # lambda
    edit_weight: float = 0.5          # e_w; c_w = (1 - e_w)/0.5
    rewrite_module_tmp: str           # "...layers.{}.mlp.fc2"
    layer_module_tmp: str
    mlp_module_tmp: str
    attn_module_tmp: str
    ln_f_module: str
    mom2_dataset: str
    mom2_n_samples: int
    mom2_dtype: str
    objective: str                    # "esd" | "ablate-dest" | "ablate-source": selects the v* cache file name
    num_edit_tokens: int = 1
    use_new_compute_z: bool = False
    sld_supervision: bool = False
    # ---- Stage 1 (v* optimisation) and evaluation knobs: carried, not used here ---------------------------------
    layer_selection: str
    fact_token: str
    v_num_grad_steps: int
    v_lr: float
    v_weight_decay: float
    clamp_norm_factor: float
    mom2_adjustment: bool
    esd_mu: Optional[Any]
    train_prompt_choice: str = "simple"
    samples_per_prompt: int = 1
    cal_text_repr_loss: bool = False
    align_obj_eos_pad: bool = False
    text_repr_loss_scale_factor: float = 0.0
    txt_img_align_scale_factor: float = 0.0
    txt_img_align_loss_metric: str = "l2"
    contrastive_text_loss: bool = False
    align_object_token: bool = False
    follow_refact: bool = True
    use_ewc: bool = False
    ewc_lambda: int = 1e4
    no_noise_loss: bool = False
    ddim_steps: Optional[int] = None
    scheduler: Optional[str] = None
    sld_type: str = "max"
    all_safe: bool = False
    add_uce_edit: bool = False
    use_sampled_noise: bool = False
    replace_repr: bool = False


@dataclass(kw_only=True)
class EMCIDHyperParams(_EditFields):
    """Stable Diffusion v1.x: one CLIP text encoder."""


@dataclass(kw_only=True)
class EMCIDXLHyperParams(_EditFields):
    """SDXL: ``layers`` / ``mom2_update_weight`` address text_encoder, the ``_2`` twins text_encoder_2."""
    layers_2: List[int]
    mom2_update_weight_2: int
