"""Hyper-parameter records of the text-encoder edit path.

Mirrors the JSON schema the reference loads with ``HyperParams.from_json`` (reference:
util/hparams.py:11-16; emcid/emcid_hparams.py:55-163 ``EMCIDHyperParams``, :166-276
``EMCIDXLHyperParams``; shipped files under hparams/*.json).  Field names, defaults and the
in-place mutability are the reference's; Stage-1-only fields are carried so the shipped JSONs load.
"""
import json
from dataclasses import dataclass, fields
from typing import Any, List, Optional


@dataclass
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


@dataclass
class EMCIDHyperParams(HyperParams):
    layers: List[int]
    layer_selection: str
    fact_token: str
    mom2_update_weight: int
    rewrite_module_tmp: str
    layer_module_tmp: str
    mlp_module_tmp: str
    attn_module_tmp: str
    ln_f_module: str
    mom2_dataset: str
    mom2_n_samples: int
    mom2_dtype: str
    v_num_grad_steps: int
    v_lr: float
    v_weight_decay: float
    clamp_norm_factor: float
    mom2_adjustment: bool
    objective: str
    esd_mu: Optional[Any]
    train_prompt_choice: str = "simple"
    use_new_compute_z: bool = False
    num_edit_tokens: int = 1
    samples_per_prompt: int = 1
    edit_weight: float = 0.5
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
    sld_supervision: bool = False
    sld_type: str = "max"
    all_safe: bool = False
    add_uce_edit: bool = False
    use_sampled_noise: bool = False
    replace_repr: bool = False


@dataclass
class EMCIDXLHyperParams(HyperParams):
    layers: List[int]
    layers_2: List[int]
    layer_selection: str
    fact_token: str
    mom2_update_weight: int
    mom2_update_weight_2: int
    rewrite_module_tmp: str
    layer_module_tmp: str
    mlp_module_tmp: str
    attn_module_tmp: str
    ln_f_module: str
    mom2_dataset: str
    mom2_n_samples: int
    mom2_dtype: str
    v_num_grad_steps: int
    v_lr: float
    v_weight_decay: float
    clamp_norm_factor: float
    mom2_adjustment: bool
    objective: str
    esd_mu: Optional[Any]
    train_prompt_choice: str = "simple"
    use_new_compute_z: bool = False
    num_edit_tokens: int = 1
    samples_per_prompt: int = 1
    edit_weight: float = 0.5
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
    sld_supervision: bool = False
    sld_type: str = "max"
    all_safe: bool = False
    add_uce_edit: bool = False
    use_sampled_noise: bool = False
    replace_repr: bool = False
