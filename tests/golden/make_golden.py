#!/usr/bin/env python3
"""Mint golden vectors by running the REAL reference (/root/reference) on CPU.

Run from the repo root, in the build container only (the GPU box has no
/root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference has no golden vectors of its own for this path, so
these fixtures pin the oracle.  The reference is imported UNMODIFIED with the
four shims of SURVEY.md §8c:
  1. stub modules for the absent `diffusers` / `torchvision` (import-only deps),
  2. a scratch cwd holding `globals.yml` + `data/ccs_filtered.json`,
  3. hparams module templates without the `text_model.` prefix (transformers 5.x),
  4. `torch.cuda.device` patched to a null context (reference crashes on CPU).
Only DATA is written to tests/golden/ (inputs + the reference's outputs).
"""
import contextlib
import io
import json
import os
import shutil
import sys
import tempfile
import types
from pathlib import Path

sys.dont_write_bytecode = True
REPO = Path(__file__).resolve().parents[2]
REF = Path("/root/reference")
OUT = REPO / "tests" / "golden"
sys.path.insert(0, str(REPO))

import numpy as np
import torch
from transformers import CLIPModel, CLIPProcessor, CLIPTokenizer, CLIPTextModel  # noqa: F401  (before stubs)
from transformers import CLIPTextModelWithProjection, CLIPVisionModelWithProjection  # noqa: F401

from emcid_amd import synthetic as syn


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    d = mod("diffusers", StableDiffusionPipeline=_Dummy, StableDiffusionXLPipeline=_Dummy,
            DDPMScheduler=_Dummy, StableDiffusionPipelineSafe=_Dummy, EulerDiscreteScheduler=_Dummy,
            UNet2DConditionModel=_Dummy, AutoencoderKL=_Dummy, DDIMScheduler=_Dummy,
            LMSDiscreteScheduler=_Dummy, PNDMScheduler=_Dummy, DPMSolverMultistepScheduler=_Dummy)
    d.models = mod("diffusers.models", UNet2DConditionModel=_Dummy)
    d.utils = mod("diffusers.utils")
    d.utils.logging = mod("diffusers.utils.logging", disable_progress_bar=lambda: None)
    d.DDPMScheduler = syn.DDPMNoiseSchedule       # Stage 1 only calls from_pretrained(...), .config.num_train_timesteps, .add_noise
    tv = mod("torchvision")
    # Stage 1's preprocess_img (compute_z.py:34-53) runs these on PIL images: stand-ins with torchvision's documented
    # behaviour for PIL inputs (torchvision is not installed in this image)
    from PIL import Image as _Image

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class Resize:
        def __init__(self, size, interpolation=None):
            self.size = size

        def __call__(self, im):
            w, h = im.size
            if (w <= h and w == self.size) or (h <= w and h == self.size):
                return im
            if w < h:
                return im.resize((self.size, int(self.size * h / w)), _Image.BILINEAR)
            return im.resize((int(self.size * w / h), self.size), _Image.BILINEAR)

    class CenterCrop:
        def __init__(self, size):
            self.size = size

        def __call__(self, im):
            w, h = im.size
            left, top = int(round((w - self.size) / 2.0)), int(round((h - self.size) / 2.0))
            return im.crop((left, top, left + self.size, top + self.size))

    class RandomHorizontalFlip:
        def __init__(self, p=0.5):
            self.p = p

        def __call__(self, im):
            return im.transpose(_Image.FLIP_LEFT_RIGHT) if torch.rand(1) < self.p else im

    class ToTensor:
        def __call__(self, im):
            return torch.from_numpy(np.asarray(im, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0)

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = mean[0], std[0]

        def __call__(self, x):
            return (x - self.mean) / self.std

    tv.transforms = mod("torchvision.transforms", Compose=Compose, Resize=Resize, CenterCrop=CenterCrop,
                        RandomHorizontalFlip=RandomHorizontalFlip, ToTensor=ToTensor, Normalize=Normalize,
                        InterpolationMode=types.SimpleNamespace(BILINEAR=0, BICUBIC=1))
    tv.datasets = mod("torchvision.datasets", ImageNet=_Dummy)
    tv.utils = mod("torchvision.utils", save_image=lambda *a, **k: None, make_grid=lambda *a, **k: None)


class _NullCudaDevice(contextlib.AbstractContextManager):
    def __init__(self, *a, **k):
        pass

    def __exit__(self, *a):
        return False


def import_reference(scratch: Path):
    _install_stubs()
    # the scratch copy of globals.yml (configuration, read relative to the cwd at import) with small Stage-1 images
    (scratch / "globals.yml").write_text((REF / "globals.yml").read_text().replace("RESOLUTION: 512", f"RESOLUTION: {STAGE1_RESOLUTION}"))
    (scratch / "data").mkdir(exist_ok=True)
    os.chdir(scratch)
    sys.path.insert(0, str(REF))
    torch.cuda.device = _NullCudaDevice
    torch.cuda.empty_cache = lambda: None
    import emcid.emcid_main as em  # noqa
    import emcid.layer_stats as ls  # noqa
    import emcid.compute_z as cz  # noqa
    from emcid.emcid_hparams import EMCIDHyperParams, EMCIDXLHyperParams
    from experiments.causal_trace import find_token_range
    return em, ls, cz, EMCIDHyperParams, EMCIDXLHyperParams, find_token_range


STAGE1_RESOLUTION = 32


class Layer427(torch.nn.Module):
    """transformers-4.27 encoder layers return a TUPLE (hidden_states,); the reference's Stage-1 hook indexes it
    (compute_z.py:353-373: cur_out[0][i, idx, :] += delta).  transformers 5.x layers return the tensor; this wrapper and
    the encoder subclass below restore the 4.27 calling convention around the unchanged layer (no arithmetic)."""

    def __init__(self, inner):
        super().__init__()
        self.inner = inner

    def forward(self, *a, **k):
        return (self.inner(*a, **k),)


def as_transformers_427(te):
    from transformers.models.clip.modeling_clip import CLIPEncoder
    from transformers.modeling_outputs import BaseModelOutput

    class CLIPEncoder427(CLIPEncoder):
        def forward(self, inputs_embeds, attention_mask=None, **kwargs):
            hidden = inputs_embeds
            for layer in self.layers:
                hidden = layer(hidden, attention_mask, **kwargs)[0]
            return BaseModelOutput(last_hidden_state=hidden)

    enc = te.encoder if hasattr(te, "encoder") else te.text_model.encoder
    enc.layers = torch.nn.ModuleList([Layer427(l) for l in enc.layers])
    enc.__class__ = CLIPEncoder427
    return te


STAGE1_CASES = {
    # the shipped SD hparams' Stage-1 settings (hparams/dest_s-200_c-1.5_ly-7-11...json), fewer steps
    "shipped": dict(hp=dict(objective="ablate-dest", cal_text_repr_loss=True, text_repr_loss_scale_factor=0.01, v_lr=0.2,
                            v_weight_decay=5e-4, clamp_norm_factor=1.5, v_num_grad_steps=10), layer=3, seed=1234, req={}),
    "ablate_source_object_token": dict(hp=dict(objective="ablate-source", cal_text_repr_loss=True, align_object_token=True,
                                               text_repr_loss_scale_factor=0.05, v_lr=0.1, v_weight_decay=1e-3,
                                               clamp_norm_factor=0.6, v_num_grad_steps=8, samples_per_prompt=2),
                                       layer=2, seed=77, req={"use_real_noise": True}),
    "eos_pad_replace": dict(hp=dict(objective="ablate-dest", cal_text_repr_loss=True, align_obj_eos_pad=True, replace_repr=True,
                                    text_repr_loss_scale_factor=0.02, v_lr=0.05, v_weight_decay=5e-4, clamp_norm_factor=1.2,
                                    v_num_grad_steps=6), layer=4, seed=5, req={}),
}


def golden_stage1(cz, EMCIDHyperParams, scratch, tag="toy_stage1"):
    """Stage 1: the REAL reference's compute_z_text_encoder (compute_z.py:315-649) on the synthetic pipe with the UNet / VAE
    stand-ins of emcid_amd/synthetic.py, the DDPM schedule stub and caller-supplied training images."""
    out, meta = {}, {"cases": {}, "resolution": STAGE1_RESOLUTION}
    for name, c in STAGE1_CASES.items():
        pipe = syn.add_diffusion(syn.build_pipe("toy", "cpu"))
        as_transformers_427(pipe.text_encoder)
        hp_d = syn.sd_hparams_dict(layers=(1, 2, 3, 4), prefix="")
        hp_d.update(c["hp"])
        hp = EMCIDHyperParams(**hp_d)
        spp = hp.samples_per_prompt
        request = {"source": "tocife" if name != "shipped" else "c0042", "dest": "a realist artist",
                   "prompts": list(syn.ARTIST_TEMPLATES), "seed_train": 2024}
        request.update(c["req"])
        imgs = syn.make_images(len(request["prompts"]) * spp, STAGE1_RESOLUTION, seed=31 + c["seed"])
        torch.manual_seed(c["seed"])
        with contextlib.redirect_stdout(io.StringIO()):
            v = cz.compute_z_text_encoder(pipe, dict(request, images=imgs), hp, c["layer"], device="cpu")
        out[f"{name}/v_star"] = v.detach().numpy()
        out[f"{name}/images"] = np.stack([np.asarray(im) for im in imgs])
        meta["cases"][name] = {"hparams": hp_d, "layer": c["layer"], "seed": c["seed"], "request": request}
    np.savez_compressed(OUT / f"{tag}.npz", **out)
    with open(OUT / f"{tag}.json", "w") as f:
        json.dump(meta, f, indent=1)
    print(f"[golden] {tag}: wrote {len(out)} arrays: " + ", ".join(f"{k}: |v*| {np.linalg.norm(out[k + '/v_star']):.4f}" for k in STAGE1_CASES))


STAGE1_MORE_CASES = {
    # hparams/dest_s-200_c-1.5_ly-11_lr-0.1_ewc-1e7_txt-align-0.01.json: EWC instead of the weight decay (compute_z.py:478-486,
    # :547-549), a synthetic Fisher file written through the reference's own Mean / CombinedStat
    "ewc": dict(hp=dict(objective="ablate-dest", cal_text_repr_loss=True, text_repr_loss_scale_factor=0.01, v_lr=0.1,
                        v_weight_decay=5e-4, use_ewc=True, ewc_lambda=1e7, clamp_norm_factor=1.5, v_num_grad_steps=12),
                layer=4, seed=4321, req={}),
    # the shipped SD settings at 50 / 100 / 150 / 200 Adam steps (v_num_grad_steps = 200 is what every shipped file sets): the
    # random draws of a shorter run are the first draws of a longer one, so the four fixtures are one trajectory sampled four times
    **{f"steps{n}": dict(hp=dict(objective="ablate-dest", cal_text_repr_loss=True, text_repr_loss_scale_factor=0.01, v_lr=0.2,
                                 v_weight_decay=5e-4, clamp_norm_factor=1.5, v_num_grad_steps=n), layer=3, seed=1234, req={})
       for n in (50, 100, 150, 200)},
}
FIM_REL = "data/fim_stats/text_encoder/ccs_filtered_stats/text_model.encoder.layers.10.mlp.fc2_float32_mean_step10_3000.npz"


def golden_stage1_more(cz, EMCIDHyperParams, scratch, tag="toy_stage1_more"):
    """More of the REAL reference's compute_z_text_encoder on the toy pipe: use_ewc (the Fisher file is written into the scratch
    cwd by the reference's own runningstats classes, the way emcid/fim_cal.py does) and the shipped step count."""
    from util.runningstats import CombinedStat, Mean
    out, meta = {}, {"cases": {}, "resolution": STAGE1_RESOLUTION, "fim_file": FIM_REL}
    g = torch.Generator().manual_seed(99)
    grads_sq = (torch.randn(40, 32, generator=g) * 1e-3) ** 2             # 40 squared-gradient samples over the toy hidden size
    stat = CombinedStat(**{"mean": Mean()})
    for chunk in grads_sq.split(10):
        stat.add(chunk)
    fim_path = Path(scratch) / FIM_REL
    fim_path.parent.mkdir(parents=True, exist_ok=True)
    stat.save(str(fim_path))
    with np.load(fim_path, allow_pickle=True) as npz:
        for k in npz.files:
            out[f"fim/{k}"] = npz[k]
    for name, c in STAGE1_MORE_CASES.items():
        pipe = syn.add_diffusion(syn.build_pipe("toy", "cpu"))
        as_transformers_427(pipe.text_encoder)
        hp_d = syn.sd_hparams_dict(layers=(1, 2, 3, 4), prefix="")
        hp_d.update(c["hp"])
        hp = EMCIDHyperParams(**hp_d)
        request = {"source": "tocife" if name == "ewc" else "c0042", "dest": "a realist artist",
                   "prompts": list(syn.ARTIST_TEMPLATES), "seed_train": 2024}
        request.update(c["req"])
        imgs = syn.make_images(len(request["prompts"]) * hp.samples_per_prompt, STAGE1_RESOLUTION, seed=31 + c["seed"])
        torch.manual_seed(c["seed"])
        with contextlib.redirect_stdout(io.StringIO()):
            v = cz.compute_z_text_encoder(pipe, dict(request, images=imgs), hp, c["layer"], device="cpu")
        out[f"{name}/v_star"] = v.detach().numpy()
        if name in ("ewc", "steps50"):
            out[f"{name}/images"] = np.stack([np.asarray(im) for im in imgs])
        meta["cases"][name] = {"hparams": hp_d, "layer": c["layer"], "seed": c["seed"], "request": request,
                               "images": name if name in ("ewc", "steps50") else "steps50"}
    np.savez_compressed(OUT / f"{tag}.npz", **out)
    with open(OUT / f"{tag}.json", "w") as f:
        json.dump(meta, f, indent=1)
    print(f"[golden] {tag}: wrote {len(out)} arrays: " + ", ".join(f"{k}: |v*| {np.linalg.norm(out[k + '/v_star']):.4f}" for k in STAGE1_MORE_CASES))


GLOBAL_STAGE1_CASES = {
    # the sld_supervision Stage 1 of a global concept (compute_z.py:77-312): "[CLS]" with images sampled from the pipeline by
    # per-prompt seeds ("max" preset), "[EOS]" with training images read from disk ("strong" preset, ablate-dest), and the esd form
    "sld_max_cls": dict(hp=dict(objective="ablate-source", sld_supervision=True, sld_type="max", v_lr=0.05, v_weight_decay=5e-4,
                                clamp_norm_factor=1.5, v_num_grad_steps=8), layer=3, seed=911, source="[CLS]", files=False),
    "sld_strong_eos_files": dict(hp=dict(objective="ablate-dest", sld_supervision=True, sld_type="strong", v_lr=0.1, v_weight_decay=1e-3,
                                         clamp_norm_factor=0.8, v_num_grad_steps=7), layer=2, seed=17, source="[EOS]", files=True),
    "esd_cls": dict(hp=dict(objective="esd", esd_mu=1.0, sld_supervision=True, sld_type="max", v_lr=0.02, v_weight_decay=5e-4,
                            clamp_norm_factor=1.2, v_num_grad_steps=6), layer=4, seed=5, source="[CLS]", files=False),
}


def golden_stage1_global(cz, EMCIDHyperParams, scratch, tag="toy_stage1_global"):
    """The REAL reference's compute_z_text_encoder_global (compute_z.py:77-312; emcid_main.py:911-918 selects it under
    ``sld_supervision``) on the synthetic pipe: UNet / VAE stand-ins, DDPM schedule stub; training images sampled by the reference
    itself through ``pipe([prompt], generator=seeded)`` or read from PNG files it is pointed at."""
    (scratch / "log").mkdir(exist_ok=True)
    out, meta = {}, {"cases": {}, "resolution": STAGE1_RESOLUTION}
    prompts = ["painting of tocife", "a photo by bamilo", "style of c0042"]
    for name, c in GLOBAL_STAGE1_CASES.items():
        pipe = syn.add_diffusion(syn.build_pipe("toy", "cpu"))
        pipe.image_resolution = STAGE1_RESOLUTION
        as_transformers_427(pipe.text_encoder)
        hp_d = syn.sd_hparams_dict(layers=(1, 2, 3, 4), prefix="")
        hp_d.update(c["hp"])
        hp = EMCIDHyperParams(**hp_d)
        request = {"source": c["source"], "dest": " ", "source_prompts": prompts, "safe_words": ["tocife", "bamilo, c0042", "artwork"],
                   "seeds": [3, 14, 15], "indices": [0, 1, 2], "prompts": [], "seed_train": 2024}
        run_req = dict(request)
        if c["files"]:
            imgs = syn.make_images(len(prompts), STAGE1_RESOLUTION, seed=31 + c["seed"])
            paths = []
            for i, im in enumerate(imgs):
                f = scratch / f"global_{name}_{i}.png"
                im.save(f)
                paths.append(str(f))
            run_req["training_img_paths"] = paths
            out[f"{name}/images"] = np.stack([np.asarray(im) for im in imgs])
        torch.manual_seed(c["seed"])
        with contextlib.redirect_stdout(io.StringIO()):
            v = cz.compute_z_text_encoder_global(pipe, run_req, hp, c["layer"], device="cpu")
        out[f"{name}/v_star"] = v.detach().numpy()
        meta["cases"][name] = {"hparams": hp_d, "layer": c["layer"], "seed": c["seed"], "request": request, "files": c["files"]}
    np.savez_compressed(OUT / f"{tag}.npz", **out)
    with open(OUT / f"{tag}.json", "w") as f:
        json.dump(meta, f, indent=1)
    print(f"[golden] {tag}: wrote {len(out)} arrays: " + ", ".join(f"{k}: |v*| {np.linalg.norm(out[k + '/v_star']):.4f}" for k in GLOBAL_STAGE1_CASES))


V1_STAGE1_CASES = {
    # the txt_img_align Stage 1 (compute_z.py:1360-1648): ablate-dest with the image-alignment term ("cos" and "l2") on top of the
    # projected-space text alignment; ablate-source without the image term, object-token alignment
    "img_align_cos": dict(hp=dict(objective="ablate-dest", cal_text_repr_loss=True, text_repr_loss_scale_factor=0.01,
                                  txt_img_align_scale_factor=0.05, txt_img_align_loss_metric="cos", v_lr=0.1, v_weight_decay=5e-4,
                                  clamp_norm_factor=1.5, v_num_grad_steps=8), layer=3, seed=404, req={"txt_img_align": True}),
    "img_align_l2_replace": dict(hp=dict(objective="ablate-dest", cal_text_repr_loss=False, replace_repr=True,
                                         txt_img_align_scale_factor=0.2, txt_img_align_loss_metric="l2", v_lr=0.05, v_weight_decay=1e-3,
                                         clamp_norm_factor=0.9, v_num_grad_steps=6), layer=2, seed=9, req={"txt_img_align": True}),
    "no_img_object_token": dict(hp=dict(objective="ablate-source", cal_text_repr_loss=True, align_object_token=True,
                                        text_repr_loss_scale_factor=0.05, txt_img_align_scale_factor=0.1, v_lr=0.1, v_weight_decay=5e-4,
                                        clamp_norm_factor=1.2, v_num_grad_steps=7), layer=4, seed=31, req={"txt_img_align": False}),
}


def golden_stage1_v1(cz, EMCIDHyperParams, scratch, tag="toy_stage1_v1"):
    """The REAL reference's compute_z_text_encoder_v1 (compute_z.py:1360-1648; emcid_main.py:919-926 selects it when
    ``txt_img_align_scale_factor != 0``) on the synthetic pipe.  Shim 5 (this function only): the reference fetches
    openai/clip-vit-large-patch14's text tower, vision tower and processor from the hub (:1376-1378, :1440) — no network here —,
    so the three ``from_pretrained`` are pointed at emcid_amd.synthetic.build_clip_towers' offline stand-ins (the text tower
    carries the pipe's own encoder weights, as the hub checkpoint does for SD-v1.x); everything else is the reference's code."""
    (scratch / "log").mkdir(exist_ok=True)
    out, meta = {}, {"cases": {}, "resolution": STAGE1_RESOLUTION, "towers": {"projection_dim": 16, "seed": 7}}
    saved = (cz.CLIPTextModelWithProjection.from_pretrained, cz.CLIPVisionModelWithProjection.from_pretrained, cz.CLIPProcessor.from_pretrained)
    try:
        for name, c in V1_STAGE1_CASES.items():
            pipe = syn.add_diffusion(syn.build_pipe("toy", "cpu"))
            pipe.image_resolution = STAGE1_RESOLUTION
            text, vision, proc = syn.build_clip_towers(pipe, image_size=STAGE1_RESOLUTION)
            as_transformers_427(text)
            cz.CLIPTextModelWithProjection.from_pretrained = staticmethod(lambda *_a, _m=text, **_k: _m)
            cz.CLIPVisionModelWithProjection.from_pretrained = staticmethod(lambda *_a, _m=vision, **_k: _m)
            cz.CLIPProcessor.from_pretrained = staticmethod(lambda *_a, _m=proc, **_k: _m)
            hp_d = syn.sd_hparams_dict(layers=(1, 2, 3, 4), prefix="text_model.")      # the hooked model is the tower WITH projection
            hp_d.update(c["hp"])
            hp = EMCIDHyperParams(**hp_d)
            request = {"source": "tocife", "dest": "a realist artist", "prompts": list(syn.ARTIST_TEMPLATES), "seed_train": 2024}
            request.update(c["req"])
            torch.manual_seed(c["seed"])
            with contextlib.redirect_stdout(io.StringIO()):
                v = cz.compute_z_text_encoder_v1(pipe, dict(request), hp, c["layer"], device="cpu")
            out[f"{name}/v_star"] = v.detach().numpy()
            meta["cases"][name] = {"hparams": hp_d, "layer": c["layer"], "seed": c["seed"], "request": request}
    finally:
        (cz.CLIPTextModelWithProjection.from_pretrained, cz.CLIPVisionModelWithProjection.from_pretrained,
         cz.CLIPProcessor.from_pretrained) = saved
    np.savez_compressed(OUT / f"{tag}.npz", **out)
    with open(OUT / f"{tag}.json", "w") as f:
        json.dump(meta, f, indent=1)
    print(f"[golden] {tag}: wrote {len(out)} arrays: " + ", ".join(f"{k}: |v*| {np.linalg.norm(out[k + '/v_star']):.4f}" for k in V1_STAGE1_CASES))


XATTN_STAGE1_CASES = {
    # safe-latent-diffusion supervision with the request's own safe words ("max" preset), the esd supervision with replace_repr,
    # and the "strong" preset over the built-in list of safety concepts
    "sld_max": dict(hp=dict(sld_supervision=True, sld_type="max", v_lr=0.05, v_weight_decay=5e-4, clamp_norm_factor=1.5,
                            v_num_grad_steps=8, samples_per_prompt=2), seed=611, req={"safe words": "tocife, bamilo"}),
    "esd_replace": dict(hp=dict(objective="esd", esd_mu=1.0, replace_repr=True, v_lr=0.02, v_weight_decay=1e-3, clamp_norm_factor=0.8,
                                v_num_grad_steps=6, samples_per_prompt=1), seed=77, req={}),
    "sld_strong_all_safe": dict(hp=dict(sld_supervision=True, sld_type="strong", all_safe=True, v_lr=0.1, v_weight_decay=5e-4,
                                        clamp_norm_factor=1.2, v_num_grad_steps=7, samples_per_prompt=2), seed=5, req={}),
}


def golden_xattn_stage1(cz, EMCIDHyperParams, scratch, tag="toy_xattn_stage1"):
    """Stage 1 of the cross-attention sibling: the REAL reference's compute_z_unet_x_kv (compute_z.py:2407-2645) on the synthetic
    pipe (UNet stand-in with the 32 attn2.to_k / to_v projections under their real names, VAE stand-in, DDPM schedule stub; the
    training images come from ``pipe(prompts, ...)`` like in the reference: the synthetic pipe draws them from the generator)."""
    (scratch / "log").mkdir(exist_ok=True)
    out, meta = {}, {"cases": {}, "resolution": STAGE1_RESOLUTION}
    for name, c in XATTN_STAGE1_CASES.items():
        pipe = syn.add_diffusion(syn.build_pipe("toy", "cpu"))
        pipe.image_resolution = STAGE1_RESOLUTION
        hp_d = syn.sd_hparams_dict(layers=(1, 2, 3, 4), prefix="")
        hp_d.update(c["hp"])
        hp = EMCIDHyperParams(**hp_d)
        request = {"source": "c0042", "dest": "a realist artist", "prompts": list(syn.ARTIST_TEMPLATES), "seed_train": 2024}
        request.update(c["req"])
        # the images the reference is about to sample (same generator, same calls), kept for runs on another device
        gen = torch.Generator("cpu").manual_seed(int(request["seed_train"]))
        src = [p.format(request["source"]) for p in request["prompts"]]
        imgs = [im for _ in range(hp.samples_per_prompt) for im in pipe(src, guidance_scale=7.5, generator=gen).images]
        torch.manual_seed(c["seed"])
        with contextlib.redirect_stdout(io.StringIO()):
            vs = cz.compute_z_unet_x_kv(pipe, dict(request), hp, "cpu")
        for ln, v in vs.items():
            out[f"{name}/v_star/{ln}"] = v.detach().numpy()
        out[f"{name}/images"] = np.stack([np.asarray(im) for im in imgs])
        meta["cases"][name] = {"hparams": hp_d, "seed": c["seed"], "request": request, "layer_names": list(vs)}
    np.savez_compressed(OUT / f"{tag}.npz", **out)
    with open(OUT / f"{tag}.json", "w") as f:
        json.dump(meta, f, indent=1)
    print(f"[golden] {tag}: wrote {len(out)} arrays; " + ", ".join(
        f"{k}: mean |v*| {np.mean([np.linalg.norm(out[a]) for a in out if a.startswith(k + '/v_star/')]):.4f}" for k in XATTN_STAGE1_CASES))


STAGE1_XL_CASES = {
    # the shipped SDXL hparams' Stage-1 settings (hparams/sdxl-dest_s-100_c-1.2_ly-8-11_ly2-26-31_lr-0.1_wd-8e-03_txt-align-0.01.json), fewer steps
    "shipped_xl": dict(hp=dict(objective="ablate-dest", cal_text_repr_loss=True, text_repr_loss_scale_factor=0.005, v_lr=0.1,
                               v_weight_decay=8e-3, clamp_norm_factor=1.2, v_num_grad_steps=8), layers=(3, 5), seed=321, req={}),
    "ablate_source_xl": dict(hp=dict(objective="ablate-source", cal_text_repr_loss=False, v_lr=0.2, v_weight_decay=1e-3,
                                     clamp_norm_factor=0.5, v_num_grad_steps=6, samples_per_prompt=2), layers=(2, 4), seed=99,
                             req={"use_real_noise": True}),
    "replace_xl": dict(hp=dict(objective="ablate-dest", cal_text_repr_loss=True, replace_repr=True, text_repr_loss_scale_factor=0.02,
                               v_lr=0.05, v_weight_decay=5e-4, clamp_norm_factor=1.2, v_num_grad_steps=5), layers=(4, 6), seed=7, req={}),
}
STAGE1_XL_PROJECTION = 40


def stage1_xl_pipe(device="cpu", wrap=True):
    """Synthetic SDXL pipe for Stage 1 of the pair: toy / toy2 encoders (the second with a projection), UNet / VAE / scheduler
    stand-ins.  ``wrap``: transformers-4.27 conventions for the REFERENCE run (tuple-returning layers, `text_model.` names)."""
    pipe = syn.add_sdxl_diffusion(syn.build_pipe("toy", device, sdxl=True, projection_dim=STAGE1_XL_PROJECTION))
    if wrap:
        as_transformers_427(pipe.text_encoder)
        as_transformers_427(pipe.text_encoder_2)
        pipe.text_encoder = _PrefixedEncoder(pipe.text_encoder)       # 5.x CLIPTextModel has no `text_model.` level; WithProjection has
    return pipe


def golden_stage1_sdxl(cz, EMCIDXLHyperParams, scratch, tag="toy_stage1_sdxl"):
    """Stage 1 of the SDXL pair: the REAL reference's compute_z_sdxl_text_encoders (compute_z.py:651-1037) on the synthetic SDXL
    pipe (UNet with added_cond_kwargs, VAE, DDPM schedule stand-ins) with caller-supplied training images."""
    (scratch / "log").mkdir(exist_ok=True)            # the reference appends its losses to log/loss_text_encoder.txt (:993)
    out, meta = {}, {"cases": {}, "resolution": STAGE1_RESOLUTION, "projection_dim": STAGE1_XL_PROJECTION}
    for name, c in STAGE1_XL_CASES.items():
        pipe = stage1_xl_pipe()
        hp_d = syn.sdxl_hparams_dict(layers=(1, 2, 3, 4), layers_2=(2, 3, 4, 5, 6), prefix="text_model.")
        hp_d.update(c["hp"])
        hp = EMCIDXLHyperParams(**hp_d)
        request = {"source": "tocife" if name != "shipped_xl" else "c0042", "dest": "a realist artist",
                   "prompts": list(syn.ARTIST_TEMPLATES), "seed_train": 2024}
        request.update(c["req"])
        imgs = syn.make_images(len(request["prompts"]) * hp.samples_per_prompt, STAGE1_RESOLUTION, seed=131 + c["seed"])
        torch.manual_seed(c["seed"])
        with contextlib.redirect_stdout(io.StringIO()):
            v1, v2 = cz.compute_z_sdxl_text_encoders(pipe, dict(request, images=imgs), hp, c["layers"], device="cpu")
        out[f"{name}/v_star"], out[f"{name}/v_star_2"] = v1.detach().numpy(), v2.detach().numpy()
        out[f"{name}/images"] = np.stack([np.asarray(im) for im in imgs])
        meta["cases"][name] = {"hparams": hp_d, "layers": list(c["layers"]), "seed": c["seed"], "request": request}
    np.savez_compressed(OUT / f"{tag}.npz", **out)
    with open(OUT / f"{tag}.json", "w") as f:
        json.dump(meta, f, indent=1)
    print(f"[golden] {tag}: wrote {len(out)} arrays: " + ", ".join(
        f"{k}: |v*| {np.linalg.norm(out[k + '/v_star']):.4f} / {np.linalg.norm(out[k + '/v_star_2']):.4f}" for k in STAGE1_XL_CASES))


def state_np(model, prefix="w/"):
    return {prefix + k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}


def record_kz(em, store):
    """Wrap the reference's K/Z assembly so its per-call outputs are recorded."""
    orig = em.get_module_input_output_at_words
    import emcid.compute_ks as cks

    def wrapped(*a, **k):
        i, o = orig(*a, **k)
        store.append((i.detach().clone(), o.detach().clone()))
        return i, o

    em.get_module_input_output_at_words = wrapped
    cks.get_module_input_output_at_words = wrapped
    return orig, cks


def unrecord_kz(em, orig, cks):
    em.get_module_input_output_at_words = orig
    cks.get_module_input_output_at_words = orig


def golden_sd(em, EMCIDHyperParams, scratch, tag, kind, n_req, layers, lam, ew, ragged, full, syllables=False,
              store_vstar=True, outliers=False, names=None, own_prompts=False):
    """Reference execute_emcid_text_encoder + apply_emcid_to_text_encoder on a synthetic pipe.  ``syllables``: bench.py's
    workload (syllable vocabulary, 3-syllable names); ``store_vstar=False``: the v* rows are a seeded function of the
    request list (syn.write_vstar_cache(seed=1, scale=0.5)), only their checksum goes into the fixture.  ``names="artist"``
    (with ``syllables="wide"``): two-word names with the first-word statistics of the reference's 1 000-artist list
    (dsets/artist_requests.py:20-46 reads data/artists/info/erased-1000artists-....txt); ``own_prompts``: every request brings
    three prompts with three words of its own in front of the subject (syn.own_prompt_requests: no shared prefixes)."""
    pipe = syn.build_pipe(kind, "cpu", syllables=syllables, outliers=outliers)
    hidden, inter = syn.ENCODER_DIMS[kind][:2]
    reqs = syn.make_requests(n_req, ragged=ragged, names=names or ("syllable" if syllables else "index"))
    if own_prompts:
        reqs = syn.own_prompt_requests(reqs)
    hp_d = syn.sd_hparams_dict(layers=layers, mom2_update_weight=lam + 1, edit_weight=0.5, mom2_n_samples=1000, prefix="")
    cache = str(scratch / f"cache_{tag}") + "/"
    stats_dir = scratch / f"stats_{tag}"
    vs = syn.write_vstar_cache(cache, reqs, hidden, seed=1, scale=0.5)
    layer_names = [hp_d["rewrite_module_tmp"].format(l) for l in layers]
    covs = syn.write_stats_cache(stats_dir, layer_names, inter, 1000, seed=2, t=max(2 * inter, 512))
    em.COV_CACHE.clear()
    hp = EMCIDHyperParams(**hp_d)
    rec = []
    orig, cks = record_kz(em, rec)
    w0 = {n: em.nethook.get_parameter(pipe.text_encoder, n + ".weight").clone() for n in layer_names}
    deltas = em.execute_emcid_text_encoder(pipe, reqs, hp, cache_name=cache, mom2_weight=lam,
                                           edit_weight=ew, verbose=False, stat_dir=str(stats_dir))
    unrecord_kz(em, orig, cks)
    for n in layer_names:  # invariant: model restored
        assert torch.equal(w0[n], em.nethook.get_parameter(pipe.text_encoder, n + ".weight"))
    # fresh hparams (reference mutates them), full apply
    hp2 = EMCIDHyperParams(**hp_d)
    em.COV_CACHE.clear()
    pipe2, orig_te = em.apply_emcid_to_text_encoder(pipe, reqs, hp2, "cpu", mom2_weight=lam, edit_weight=ew,
                                                    return_orig_text_encoder=True, cache_name=cache,
                                                    stats_dir=str(stats_dir), verbose=False)
    assert hp2.mom2_update_weight == lam  # in-place mutation quirk
    out = {"vstar": vs} if store_vstar else {"vstar_sum": np.array(vs.astype(np.float64).sum()),
                                             "vstar_row0": vs[0]}
    meta = {"kind": kind, "requests": reqs if store_vstar else None, "n_requests": n_req, "syllables": syllables, "outliers": outliers,
            "names": names or ("syllable" if syllables else "index"), "own_prompts": own_prompts,
            "hparams": hp_d, "layers": list(layers), "lam": lam, "ew": ew, "layer_names": layer_names,
            "stats": {"seed": 2, "t": max(2 * inter, 512), "n_samples": 1000}, "vstar": {"seed": 1, "scale": 0.5}}
    g = torch.Generator().manual_seed(123)
    probe = torch.randn(inter, 8, generator=g, dtype=torch.float64)
    for li, n in enumerate(layer_names):
        adj_k, resid = deltas[n + ".weight"]
        k_in, _ = rec[2 * li]
        _, z_out = rec[2 * li + 1]
        w_new = em.nethook.get_parameter(pipe2.text_encoder, n + ".weight")
        dw = (w_new.double() - w0[n].double())
        if full:
            out[f"cov/{li}"] = covs[n].astype(np.float32)
            out[f"K/{li}"] = k_in.numpy()
            out[f"Zc/{li}"] = z_out.numpy()
            out[f"adj_k/{li}"] = adj_k.numpy()
            out[f"resid/{li}"] = resid.numpy()
            out[f"w_final/{li}"] = w_new.numpy()
            out[f"w_orig/{li}"] = w0[n].numpy()
        else:
            out[f"K_probe/{li}"] = (k_in.double() @ probe).numpy()
            out[f"Zc_rownorm/{li}"] = z_out.double().norm(dim=1).numpy()
            out[f"adjk_probe/{li}"] = (probe.t() @ adj_k).numpy()          # (8, N)
            out[f"resid_fro/{li}"] = np.array(resid.norm().item())
            out[f"dw_probe/{li}"] = (dw @ probe).numpy()                   # (h, 8)
            out[f"dw_fro/{li}"] = np.array(dw.norm().item())
            out[f"dw_rownorm/{li}"] = dw.norm(dim=1).numpy()
            out[f"dw_maxabs/{li}"] = np.array(dw.abs().max().item())
    if full:
        out.update(state_np(orig_te))
    np.savez_compressed(OUT / f"{tag}.npz", **out)
    with open(OUT / f"{tag}.json", "w") as f:
        json.dump(meta, f, indent=1)
    print(f"[golden] {tag}: wrote {len(out)} arrays")


MULTI_TOKEN = dict(k=3, n_req=4, layers=(1, 2, 3, 4), lam=60, ew=0.55, seed=4321,
                   hp=dict(objective="ablate-dest", cal_text_repr_loss=True, text_repr_loss_scale_factor=0.02, v_lr=0.1,
                           v_weight_decay=5e-4, clamp_norm_factor=0.8, v_num_grad_steps=6, use_new_compute_z=True, num_edit_tokens=3))


def golden_multi_token(em, EMCIDHyperParams, scratch, tag="toy_multi_token"):
    """``use_new_compute_z`` with ``num_edit_tokens = 3``, end to end through the REAL reference: an empty v* cache, so
    execute_emcid_text_encoder runs compute_z_text_encoder_v2 per request (compute_z.py:1041-1357; files of shape (3, hidden),
    emcid_main.py:946-957), then the layer loop on (N, 3, .) key / value rows flattened "rq num" (:993-1014), then the
    apply.  UNet / VAE / DDPM stand-ins and 4.27 layer convention as for toy_stage1; the wrapper puts the layers' own modules
    one level down (``.inner``), which the REFERENCE run's rewrite_module_tmp spells out (data, not code)."""
    c = MULTI_TOKEN
    pipe = syn.add_diffusion(syn.build_pipe("toy", "cpu"))
    w_names = [f"encoder.layers.{l}.mlp.fc2" for l in c["layers"]]
    w0 = {n: em.nethook.get_parameter(pipe.text_encoder, n + ".weight").detach().clone() for n in w_names}
    state0 = {k: v.copy() for k, v in state_np(pipe.text_encoder).items()}      # (state_np's arrays alias the live weights)
    as_transformers_427(pipe.text_encoder)
    hidden, inter = syn.ENCODER_DIMS["toy"][:2]
    reqs = syn.make_requests(c["n_req"], ragged=True)
    hp_d = syn.sd_hparams_dict(layers=c["layers"], mom2_update_weight=c["lam"] + 1, edit_weight=0.5, mom2_n_samples=1000, prefix="")
    hp_d.update(c["hp"])
    ref_d = dict(hp_d, rewrite_module_tmp="encoder.layers.{}.inner.mlp.fc2")
    imgs = [syn.make_images(len(r["prompts"]) * hp_d.get("samples_per_prompt", 1), STAGE1_RESOLUTION, seed=500 + i)
            for i, r in enumerate(reqs)]
    run_reqs = [dict(r, images=im) for r, im in zip(reqs, imgs)]
    cache = str(scratch / f"cache_{tag}") + "/"
    stats_dir = scratch / f"stats_{tag}"
    ref_names = [ref_d["rewrite_module_tmp"].format(l) for l in c["layers"]]
    covs = syn.write_stats_cache(stats_dir, ref_names, inter, 1000, seed=2, t=max(2 * inter, 512))
    em.COV_CACHE.clear()
    rec = []
    orig, cks = record_kz(em, rec)
    torch.manual_seed(c["seed"])
    with contextlib.redirect_stdout(io.StringIO()):
        deltas = em.execute_emcid_text_encoder(pipe, run_reqs, EMCIDHyperParams(**ref_d), cache_name=cache, mom2_weight=c["lam"],
                                               edit_weight=c["ew"], verbose=False, stat_dir=str(stats_dir))
    unrecord_kz(em, orig, cks)
    em.COV_CACHE.clear()
    with contextlib.redirect_stdout(io.StringIO()):
        pipe2, _ = em.apply_emcid_to_text_encoder(pipe, run_reqs, EMCIDHyperParams(**ref_d), "cpu", mom2_weight=c["lam"],
                                                  edit_weight=c["ew"], cache_name=cache, stats_dir=str(stats_dir), verbose=False)
    out = dict(state0)
    for i, r in enumerate(reqs):
        with np.load(cache + f"source_{r['source']}_dest_{r['dest']}.npz") as z:
            out[f"vstar/{i}"] = z["v_star"]
        assert out[f"vstar/{i}"].shape == (c["k"], hidden)
        out[f"images/{i}"] = np.stack([np.asarray(im) for im in imgs[i]])
    for li, (n, rn) in enumerate(zip(w_names, ref_names)):
        adj_k, resid = deltas[rn + ".weight"]
        k_in, _ = rec[2 * li]
        _, z_out = rec[2 * li + 1]
        out[f"cov/{li}"] = covs[rn].astype(np.float32)
        out[f"K/{li}"] = k_in.numpy()                 # (N, k, d)
        out[f"Zc/{li}"] = z_out.numpy()               # (N, k, h)
        out[f"adj_k/{li}"] = adj_k.numpy()            # (d, N k)
        out[f"resid/{li}"] = resid.numpy()            # (h, N k)
        out[f"w_orig/{li}"] = w0[n].numpy()
        out[f"w_final/{li}"] = em.nethook.get_parameter(pipe2.text_encoder, rn + ".weight").detach().numpy()
    meta = {"kind": "toy", "requests": reqs, "hparams": hp_d, "layers": list(c["layers"]), "lam": c["lam"], "ew": c["ew"],
            "seed": c["seed"], "k": c["k"], "layer_names": w_names, "resolution": STAGE1_RESOLUTION,
            "stats": {"seed": 2, "t": max(2 * inter, 512), "n_samples": 1000}}
    np.savez_compressed(OUT / f"{tag}.npz", **out)
    with open(OUT / f"{tag}.json", "w") as f:
        json.dump(meta, f, indent=1)
    print(f"[golden] {tag}: wrote {len(out)} arrays; |v*| " + ", ".join(f"{np.linalg.norm(out[f'vstar/{i}']):.4f}" for i in range(len(reqs))))


def golden_sdxl(em, EMCIDXLHyperParams, scratch, tag="toy_sdxl"):
    pipe = syn.build_pipe("toy", "cpu", sdxl=True)
    reqs = syn.make_requests(6)
    layers, layers_2 = (2, 3), (3, 4, 5)
    hp_d = syn.sdxl_hparams_dict(layers=layers, layers_2=layers_2, mom2_update_weight=40, mom2_update_weight_2=90,
                                 mom2_n_samples=1000, prefix="")
    cache = str(scratch / f"cache_{tag}") + "/"
    sd1, sd2 = scratch / f"stats1_{tag}", scratch / f"stats2_{tag}"
    h1, i1 = syn.ENCODER_DIMS["toy"][:2]
    h2, i2 = syn.ENCODER_DIMS["toy2"][:2]
    vs1 = syn.write_vstar_cache(cache, reqs, h1, seed=1, scale=0.5)
    vs2 = syn.write_vstar_cache(cache, reqs, h2, seed=5, scale=0.5, suffix="_2")
    n1 = [hp_d["rewrite_module_tmp"].format(l) for l in layers]
    n2 = [hp_d["rewrite_module_tmp"].format(l) for l in layers_2]
    c1 = syn.write_stats_cache(sd1, n1, i1, 1000, seed=2, t=512)
    c2 = syn.write_stats_cache(sd2, n2, i2, 1000, seed=7, t=512)
    em.COV_CACHE.clear()
    w0_1 = {n: em.nethook.get_parameter(pipe.text_encoder, n + ".weight").clone() for n in n1}
    w0_2 = {n: em.nethook.get_parameter(pipe.text_encoder_2, n + ".weight").clone() for n in n2}
    hp = EMCIDXLHyperParams(**hp_d)
    pipe, o1, o2 = em.apply_emcid_to_sdxl_text_encoders(
        pipe, reqs, hp, "cpu", mom2_weight=50, mom2_weight_2=100, edit_weight=0.6,
        return_orig_text_encoder=True, cache_name=cache, stat_dir=str(sd1), stat_dir_2=str(sd2), verbose=False)
    out = {"vstar": vs1, "vstar_2": vs2}
    for li, n in enumerate(n1):
        out[f"cov/{li}"] = c1[n].astype(np.float32)
        out[f"w_orig/{li}"] = w0_1[n].numpy()
        out[f"w_final/{li}"] = em.nethook.get_parameter(pipe.text_encoder, n + ".weight").numpy()
    for li, n in enumerate(n2):
        out[f"cov_2/{li}"] = c2[n].astype(np.float32)
        out[f"w_orig_2/{li}"] = w0_2[n].numpy()
        out[f"w_final_2/{li}"] = em.nethook.get_parameter(pipe.text_encoder_2, n + ".weight").numpy()
    out.update(state_np(o1, "w1/"))
    out.update(state_np(o2, "w2/"))
    np.savez_compressed(OUT / f"{tag}.npz", **out)
    with open(OUT / f"{tag}.json", "w") as f:
        json.dump({"requests": reqs, "hparams": hp_d, "layers": layers, "layers_2": layers_2,
                   "mom2_weight": 50, "mom2_weight_2": 100, "edit_weight": 0.6,
                   "layer_names": n1, "layer_names_2": n2}, f, indent=1)
    print(f"[golden] {tag}: wrote {len(out)} arrays")


def golden_sdxl_real(em, EMCIDXLHyperParams, scratch, tag="real_sdxl_summary", n_req=300):
    """BASELINE config 4 at real dimensions: apply_emcid_to_sdxl_text_encoders of the REAL reference on the synthetic
    SDXL pair (TE1 768/3072/12L layers 8-10, TE2 1280/5120/32L layers 26-30), bench vocabulary, N concepts; summaries of
    the final dW of both encoders (TE2 carries the reference's double apply)."""
    pipe = syn.build_pipe("sdxl", "cpu", sdxl=True, syllables=True)
    reqs = syn.make_requests(n_req, names="syllable")
    layers, layers_2 = (8, 9, 10), (26, 27, 28, 29, 30)
    hp_d = syn.sdxl_hparams_dict(layers=layers, layers_2=layers_2, mom2_update_weight=4000, mom2_update_weight_2=10000,
                                 mom2_n_samples=1000, prefix="")
    cache = str(scratch / f"cache_{tag}") + "/"
    sd1, sd2 = scratch / f"stats1_{tag}", scratch / f"stats2_{tag}"
    h1, i1 = syn.ENCODER_DIMS["sdxl-te1"][:2]
    h2, i2 = syn.ENCODER_DIMS["sdxl-te2"][:2]
    vs1 = syn.write_vstar_cache(cache, reqs, h1, seed=1, scale=0.5)
    vs2 = syn.write_vstar_cache(cache, reqs, h2, seed=5, scale=0.5, suffix="_2")
    n1 = [hp_d["rewrite_module_tmp"].format(l) for l in layers]
    n2 = [hp_d["rewrite_module_tmp"].format(l) for l in layers_2]
    syn.write_stats_cache(sd1, n1, i1, 1000, seed=2, t=2 * i1)
    syn.write_stats_cache(sd2, n2, i2, 1000, seed=7, t=2 * i2)
    em.COV_CACHE.clear()
    w0_1 = {n: em.nethook.get_parameter(pipe.text_encoder, n + ".weight").clone() for n in n1}
    w0_2 = {n: em.nethook.get_parameter(pipe.text_encoder_2, n + ".weight").clone() for n in n2}
    hp = EMCIDXLHyperParams(**hp_d)
    pipe, _, _ = em.apply_emcid_to_sdxl_text_encoders(pipe, reqs, hp, "cpu", cache_name=cache, stat_dir=str(sd1),
                                                      stat_dir_2=str(sd2), verbose=False)
    out = {"vstar_sum": np.array(vs1.astype(np.float64).sum()), "vstar_2_sum": np.array(vs2.astype(np.float64).sum())}
    g = torch.Generator().manual_seed(123)
    for sfx, names, w0, te, inter in (("", n1, w0_1, pipe.text_encoder, i1), ("_2", n2, w0_2, pipe.text_encoder_2, i2)):
        probe = torch.randn(inter, 8, generator=g, dtype=torch.float64)
        out[f"probe{sfx}"] = probe.numpy()
        for li, n in enumerate(names):
            dw = em.nethook.get_parameter(te, n + ".weight").double() - w0[n].double()
            out[f"dw_probe{sfx}/{li}"] = (dw @ probe).numpy()
            out[f"dw_fro{sfx}/{li}"] = np.array(dw.norm().item())
            out[f"dw_rownorm{sfx}/{li}"] = dw.norm(dim=1).numpy()
            out[f"dw_maxabs{sfx}/{li}"] = np.array(dw.abs().max().item())
    np.savez_compressed(OUT / f"{tag}.npz", **out)
    with open(OUT / f"{tag}.json", "w") as f:
        json.dump({"n_requests": n_req, "hparams": hp_d, "layers": layers, "layers_2": layers_2, "layer_names": n1,
                   "layer_names_2": n2, "stats": {"seed": 2, "seed_2": 7, "n_samples": 1000},
                   "vstar": {"seed": 1, "seed_2": 5, "scale": 0.5}}, f, indent=1)
    print(f"[golden] {tag}: wrote {len(out)} arrays")


def golden_stage0_real(ls, scratch, tag="real_stage0_summary", n_captions=20000, layers=(2, 7)):
    """BASELINE config 5 at real dimensions: the REAL reference's layer_stats_text_encoder over n_captions synthetic
    captions (SD-v1.4 dims, d = 3072), two layers; summaries of C = mom2 / count."""
    from tqdm import tqdm
    syn.write_captions(scratch / "data" / "ccs_filtered.json", n_captions, seed=2)
    pipe = syn.build_pipe("sd-v1.4", "cpu")
    d = syn.ENCODER_DIMS["sd-v1.4"][1]
    g = torch.Generator().manual_seed(321)
    probe = torch.randn(d, 8, generator=g, dtype=torch.float64)
    out = {"probe": probe.numpy()}
    names = [f"encoder.layers.{l}.mlp.fc2" for l in layers]
    for li, ln in enumerate(names):
        stat = ls.layer_stats_text_encoder(pipe.text_encoder, pipe.tokenizer, ln, str(scratch / "stats0_real"),
                                           "ccs_filtered", ["mom2"], sample_size=n_captions, precision="float32",
                                           batch_tokens=3 * 1024, progress=tqdm)
        mom2 = stat.mom2.mom2.double()
        cnt = int(stat.mom2.count)
        C = mom2 / cnt
        out[f"count/{li}"] = np.array(cnt)
        out[f"trace/{li}"] = np.array(C.diagonal().sum().item())
        out[f"fro/{li}"] = np.array(C.norm().item())
        out[f"diag/{li}"] = C.diagonal().numpy()
        out[f"C_probe/{li}"] = (C @ probe).numpy()
    np.savez_compressed(OUT / f"{tag}.npz", **out)
    with open(OUT / f"{tag}.json", "w") as f:
        json.dump({"kind": "sd-v1.4", "n_captions": n_captions, "captions": {"seed": 2}, "layer_names": names,
                   "sample_size": n_captions, "batch_tokens": 3 * 1024}, f, indent=1)
    print(f"[golden] {tag}: wrote {len(out)} arrays")


def golden_toy_extras(cz, ls, scratch, tag="toy_extras"):
    """Two reference behaviours no shipped hparams file exercises: get_module_input_output_at_words with
    num_fact_token = 3 (compute_z.py:2329-2382) and layer_stats_text_encoder with precision = float64 (layer_stats.py:161,218)."""
    from tqdm import tqdm
    pipe = syn.build_pipe("toy", "cpu")
    reqs = syn.make_requests(7, ragged=True)
    mod = "encoder.layers.3.mlp.fc2"
    out = {}
    for k in (2, 3):
        i, o = cz.get_module_input_output_at_words(pipe.text_encoder, pipe.tokenizer, reqs, mod, num_fact_token=k)
        out[f"K{k}"], out[f"Z{k}"] = i.detach().numpy(), o.detach().numpy()
    caps = syn.write_captions(scratch / "data" / "ccs_filtered.json", 300, seed=6)
    ln = "encoder.layers.2.mlp.fc2"
    stat = ls.layer_stats_text_encoder(pipe.text_encoder, pipe.tokenizer, ln, str(scratch / "stats64"), "ccs_filtered",
                                       ["mom2"], sample_size=200, precision="float64", batch_tokens=600, progress=tqdm)
    out["mom2_f64"] = stat.mom2.mom2.numpy()
    out["count_f64"] = np.array(stat.mom2.count)
    assert out["mom2_f64"].dtype == np.float64
    f = syn.stats_file(scratch / "stats64", ln, 200, precision="float64", batch_tokens=600)
    with np.load(f) as z:
        assert z["mom2.mom2"].dtype == np.float64
    np.savez_compressed(OUT / f"{tag}.npz", **out)
    with open(OUT / f"{tag}.json", "w") as fh:
        json.dump({"kind": "toy", "requests": reqs, "module": mod, "captions": caps, "stats_layer": ln, "sample_size": 200,
                   "batch_tokens": 600}, fh)
    print(f"[golden] {tag}: wrote {len(out)} arrays; K3 {out['K3'].shape}")


def golden_stage0(ls, scratch, tag="toy_stage0"):
    from tqdm import tqdm
    caps = syn.write_captions(scratch / "data" / "ccs_filtered.json", 700, seed=2)
    pipe = syn.build_pipe("toy", "cpu")
    out = {}
    layer_names = ["encoder.layers.1.mlp.fc2", "encoder.layers.4.mlp.fc2"]
    for li, ln in enumerate(layer_names):
        stat = ls.layer_stats_text_encoder(pipe.text_encoder, pipe.tokenizer, ln, str(scratch / "stats0"),
                                           "ccs_filtered", ["mom2"], sample_size=500, precision="float32",
                                           batch_tokens=600, progress=tqdm)
        out[f"mom2/{li}"] = stat.mom2.mom2.numpy()
        out[f"count/{li}"] = np.array(stat.mom2.count)
        # the npz the reference wrote (format pin)
        f = syn.stats_file(scratch / "stats0", ln, 500, batch_tokens=600)
        with np.load(f) as z:
            out[f"npz_keys/{li}"] = np.array(sorted(z.files))
            assert int(z["sample_size"]) == 500
    np.savez_compressed(OUT / f"{tag}.npz", **out)
    with open(OUT / f"{tag}.json", "w") as f:
        json.dump({"captions": caps, "layer_names": layer_names, "sample_size": 500, "batch_tokens": 600,
                   "kind": "toy"}, f)
    print(f"[golden] {tag}: wrote {len(out)} arrays")


def golden_instruction(em, EMCIDHyperParams, scratch, tag="config1_van_gogh"):
    """BASELINE config 1: the reference's own instruction file test_examples/erasing_van_gogh_style.json with its shipped
    hparams file, applied by the REAL reference to the synthetic SD-v1.4-dimension encoder (N = 1 request, 3 prompts).
    The instruction and hparams JSON are stored verbatim in the fixture (they are data); summaries of dW as in
    real_sd_summary."""
    ins = json.load(open(REF / "test_examples" / "erasing_van_gogh_style.json"))
    hp_file = json.load(open(REF / "hparams" / f"{ins['hparams']}.json"))
    kind = "sd-v1.4"
    pipe = syn.build_pipe(kind, "cpu")
    hidden, inter = syn.ENCODER_DIMS[kind][:2]
    hp_d = dict(hp_file)
    for k in ("rewrite_module_tmp", "layer_module_tmp", "mlp_module_tmp", "attn_module_tmp", "ln_f_module"):
        hp_d[k] = hp_d[k].replace("text_model.", "")          # shim 3 (transformers 5.x names), for the reference only
    cache = str(scratch / "cache" / ins["hparams"]) + "/"
    stats_dir = scratch / f"stats_{tag}"
    vs = syn.write_vstar_cache(cache, ins["requests"], hidden, seed=1, scale=0.5)
    names = [hp_d["rewrite_module_tmp"].format(l) for l in hp_d["layers"]]
    syn.write_stats_cache(stats_dir, names, inter, hp_d["mom2_n_samples"], seed=2, t=2 * inter)
    em.COV_CACHE.clear()
    hp = EMCIDHyperParams(**hp_d)
    hp.mom2_update_weight, hp.edit_weight = ins["mom2_weight"], ins["edit_weight"]          # set_weights, emcid_test.py:924-930
    w0 = {n: em.nethook.get_parameter(pipe.text_encoder, n + ".weight").clone() for n in names}
    em.apply_emcid_to_text_encoder(pipe, ins["requests"], hp, "cpu", cache_name=cache, stats_dir=str(stats_dir), verbose=False)
    g = torch.Generator().manual_seed(123)
    probe = torch.randn(inter, 8, generator=g, dtype=torch.float64)
    out = {"vstar": vs}
    for li, n in enumerate(names):
        dw = em.nethook.get_parameter(pipe.text_encoder, n + ".weight").double() - w0[n].double()
        out[f"dw_probe/{li}"] = (dw @ probe).numpy()
        out[f"dw_fro/{li}"] = np.array(dw.norm().item())
        out[f"dw_rownorm/{li}"] = dw.norm(dim=1).numpy()
        out[f"dw_maxabs/{li}"] = np.array(dw.abs().max().item())
    np.savez_compressed(OUT / f"{tag}.npz", **out)
    with open(OUT / f"{tag}.json", "w") as f:
        json.dump({"kind": kind, "instruction": ins, "hparams_file": hp_file, "layer_names": names,
                   "stats": {"seed": 2, "t": 2 * inter}, "vstar": {"seed": 1, "scale": 0.5}}, f, indent=1)
    print(f"[golden] {tag}: wrote {len(out)} arrays")


def golden_cal_insert(em, EMCIDHyperParams, scratch, tag="toy_cal_insert"):
    """Reference cal_insert_deltas (emcid_main.py:1969-2052): the layer loop for caller-supplied targets; it reads the
    statistics from the module-level STATS_DIR and leaves the model edited."""
    pipe = syn.build_pipe("toy", "cpu")
    hidden, inter = syn.ENCODER_DIMS["toy"][:2]
    reqs = syn.make_requests(5)
    layers = (2, 3)
    hp_d = syn.sd_hparams_dict(layers=layers, mom2_update_weight=35, edit_weight=0.7, mom2_n_samples=1000, prefix="")
    names = [hp_d["rewrite_module_tmp"].format(l) for l in layers]
    covs = syn.write_stats_cache(scratch / "data" / "stats", names, inter, 1000, seed=13, t=512)
    em.COV_CACHE.clear()
    g = torch.Generator().manual_seed(77)
    zs = torch.randn(hidden, len(reqs), generator=g) * 0.5
    hp = EMCIDHyperParams(**hp_d)
    weights = {n + ".weight": em.nethook.get_parameter(pipe.text_encoder, n + ".weight") for n in names}
    w0 = {k: v.clone() for k, v in weights.items()}
    deltas = em.cal_insert_deltas(pipe, weights, hp, reqs, zs, verbose=False)
    out = {"zs": zs.numpy()}
    for li, n in enumerate(names):
        adj_k, resid = deltas[n + ".weight"]
        out[f"cov/{li}"] = covs[n].astype(np.float32)
        out[f"adj_k/{li}"] = adj_k.numpy()
        out[f"resid/{li}"] = resid.numpy()
        out[f"w_orig/{li}"] = w0[n + ".weight"].numpy()
        out[f"w_after/{li}"] = weights[n + ".weight"].detach().numpy().copy()    # NOT restored by the reference
    for k, v in w0.items():
        weights[k][...] = v
    out.update(state_np(pipe.text_encoder))
    np.savez_compressed(OUT / f"{tag}.npz", **out)
    with open(OUT / f"{tag}.json", "w") as f:
        json.dump({"kind": "toy", "requests": reqs, "hparams": hp_d, "layers": list(layers), "layer_names": names}, f, indent=1)
    print(f"[golden] {tag}: wrote {len(out)} arrays")


def golden_xattn(em, EMCIDHyperParams, scratch, tag="toy_xattn"):
    """Reference execute_emcid_cross_attn + apply_emcid_to_cross_attn (emcid_main.py:314-548) on a synthetic pipe whose
    `unet` is the module tree of cross-attention K/V projections (emcid_amd/synthetic.py: SyntheticUNet)."""
    pipe = syn.add_unet(syn.build_pipe("toy", "cpu"), "toy")
    hidden = syn.ENCODER_DIMS["toy"][0]
    reqs = syn.make_requests(6)
    lam, ew = 30, 0.6
    hp_d = syn.sd_hparams_dict(layers=(1,), mom2_update_weight=lam + 1, edit_weight=0.5, mom2_n_samples=1000, prefix="")
    names = em.get_all_cross_attn_kv_layer_names(pipe)
    mods = dict(pipe.unet.named_modules())
    dims = {n: mods[n].out_features for n in names}
    cache = str(scratch / f"cache_{tag}") + "/"
    vs = syn.write_xattn_vstar_cache(cache, reqs, dims, seed=4, scale=0.5)
    # the reference reads these from its module-level STATS_DIR ("data/stats" under the scratch cwd)
    covs = syn.write_stats_cache(scratch / "data" / "stats", names, hidden, 1000, seed=9, t=512, model_name="unet")
    em.COV_CACHE.clear()
    rec = {}
    orig = em.get_layers_input_output_at_words_cross_attn

    def wrapped(*a, **k):
        i, o = orig(*a, **k)
        rec["K"] = {n: v.detach().clone() for n, v in i.items()}
        rec["Zc"] = {n: v.detach().clone() for n, v in o.items()}
        return i, o

    em.get_layers_input_output_at_words_cross_attn = wrapped
    w0 = {n: em.nethook.get_parameter(pipe.unet, n + ".weight").clone() for n in names}
    hp = EMCIDHyperParams(**hp_d)
    deltas = em.execute_emcid_cross_attn(pipe, reqs, hp, cache_name=cache, mom2_weight=lam, edit_weight=ew, verbose=False)
    for n in names:   # invariant: UNet restored
        assert torch.equal(w0[n], em.nethook.get_parameter(pipe.unet, n + ".weight"))
    assert hp.mom2_update_weight == lam and hp.edit_weight == ew      # in-place mutation, as in the text-encoder path
    hp2 = EMCIDHyperParams(**hp_d)
    em.COV_CACHE.clear()
    pipe2, orig_unet = em.apply_emcid_to_cross_attn(pipe, reqs, hp2, "cpu", mom2_weight=lam, edit_weight=ew,
                                                    return_orig_text_model=True, cache_name=cache)
    em.get_layers_input_output_at_words_cross_attn = orig
    out = {}
    for li, n in enumerate(names):
        adj_k, resid = deltas[n + ".weight"]
        out[f"vstar/{li}"] = vs[n]
        out[f"cov/{li}"] = covs[n].astype(np.float32)
        out[f"K/{li}"] = rec["K"][n].numpy()
        out[f"Zc/{li}"] = rec["Zc"][n].numpy()
        out[f"adj_k/{li}"] = adj_k.numpy()
        out[f"resid/{li}"] = resid.numpy()
        out[f"w_orig/{li}"] = w0[n].numpy()
        out[f"w_final/{li}"] = em.nethook.get_parameter(pipe2.unet, n + ".weight").numpy()
        assert torch.equal(em.nethook.get_parameter(orig_unet, n + ".weight"), w0[n])
    out.update(state_np(pipe.text_encoder, "te/"))
    # Stage 0 of this path: the reference's layer_stats_cross_attn_kv on the captions golden_stage0 wrote
    from tqdm import tqdm
    import emcid.layer_stats as ls
    stat = ls.layer_stats_cross_attn_kv(pipe, names[3], str(scratch / "xstats0"), "ccs_filtered", ["mom2"],
                                        sample_size=300, precision="float32", batch_tokens=600, progress=tqdm)
    out["stage0/mom2"] = stat.mom2.mom2.numpy()
    out["stage0/count"] = np.array(stat.mom2.count)
    np.savez_compressed(OUT / f"{tag}.npz", **out)
    with open(OUT / f"{tag}.json", "w") as f:
        json.dump({"kind": "toy", "requests": reqs, "hparams": hp_d, "lam": lam, "ew": ew, "layer_names": names,
                   "unet_seed": 11, "stage0": {"layer": names[3], "sample_size": 300, "batch_tokens": 600,
                                               "captions_from": "toy_stage0.json"}}, f, indent=1)
    print(f"[golden] {tag}: wrote {len(out)} arrays, {len(names)} projections")


class _PrefixedEncoder(torch.nn.Module):
    """transformers-4.x module names (`text_model.encoder...`, hard-coded at uce_train.py:45) for a 5.x CLIPTextModel
    (SURVEY.md 8c shim 3: wrap the model; no arithmetic)."""

    def __init__(self, te):
        super().__init__()
        self.text_model = te
        self.config = te.config

    def forward(self, *a, **k):
        return self.text_model(*a, **k)


UCE_CASES = {
    # text-encoder fc2 variant (uce_train.py:31-213)
    "te_tensor": dict(kind="te", layer_to_edit=3, technique="tensor", retain=["painting", "a photo of the artist"],
                      lamb=0.1, erase_scale=0.1, preserve_scale=0.1),
    "te_replace": dict(kind="te", layer_to_edit=4, technique="replace", retain=None, lamb=0.5, erase_scale=0.2,
                       preserve_scale=0.3),
    # cross-attention K/V variant (uce_train.py:216-416)
    "ca_tensor": dict(kind="ca", layers_to_edit=None, with_to_k=True, technique="tensor",
                      retain=["painting", "a photo of the artist"], lamb=0.1, erase_scale=0.1, preserve_scale=0.1),
    "ca_replace_subset": dict(kind="ca", layers_to_edit=[0, 2, 3, 9, 17, 31], with_to_k=False, technique="replace",
                              retain=None, lamb=0.5, erase_scale=1.0, preserve_scale=0.1),
}
UCE_OLD = ["tocife", "gefeti vonibo", "famous sketch", "bako"]
UCE_NEW = ["a realist artist", "", "drawing", "landscape painting with a portrait"]


def golden_uce(scratch, tag="toy_uce"):
    """Reference edit_text_encoder_uce / edit_model_uce (emcid/uce_train.py) on the synthetic pipe + SyntheticUNet."""
    import emcid.uce_train as uce
    out, meta = {}, {"old": UCE_OLD, "new": UCE_NEW, "cases": UCE_CASES, "unet_seed": 11}
    out.update(state_np(syn.build_pipe("toy", "cpu").text_encoder, "te/"))
    out.update(state_np(syn.add_unet(syn.build_pipe("toy", "cpu"), "toy").unet, "unet/"))
    for name, c in UCE_CASES.items():
        pipe = syn.add_unet(syn.build_pipe("toy", "cpu"), "toy")
        kw = dict(lamb=c["lamb"], erase_scale=c["erase_scale"], preserve_scale=c["preserve_scale"], technique=c["technique"])
        with contextlib.redirect_stdout(io.StringIO()):
            if c["kind"] == "te":
                te = pipe.text_encoder
                pipe.text_encoder = _PrefixedEncoder(te)
                fc2 = te.encoder.layers[c["layer_to_edit"]].mlp.fc2
                out[f"{name}/w_orig"] = fc2.weight.detach().numpy().copy()
                uce.edit_text_encoder_uce(pipe, UCE_OLD, UCE_NEW, c["retain"], layer_to_edit=c["layer_to_edit"], **kw)
                out[f"{name}/w_final"] = fc2.weight.detach().numpy().copy()
            else:
                names = [n for n, m in pipe.unet.named_modules() if n.endswith((".to_k", ".to_v"))]
                w0 = {n: dict(pipe.unet.named_modules())[n].weight.detach().numpy().copy() for n in names}
                uce.edit_model_uce(pipe, UCE_OLD, UCE_NEW, c["retain"], layers_to_edit=c["layers_to_edit"],
                                   with_to_k=c["with_to_k"], **kw)
                mods = dict(pipe.unet.named_modules())
                changed = []
                for n in names:
                    w1 = mods[n].weight.detach().numpy()
                    if not np.array_equal(w1, w0[n]):
                        changed.append(n)
                        out[f"{name}/w_final/{n}"] = w1.copy()
                meta.setdefault("changed", {})[name] = changed
    np.savez_compressed(OUT / f"{tag}.npz", **out)
    with open(OUT / f"{tag}.json", "w") as f:
        json.dump(meta, f, indent=1)
    print(f"[golden] {tag}: wrote {len(out)} arrays; changed projections: " +
          ", ".join(f"{k}={len(v)}" for k, v in meta.get("changed", {}).items()))


def golden_token_ranges(find_token_range, tag="token_ranges"):
    tok = syn.build_tokenizer()
    cases = [
        ("painting by c0042", "c0042"), ("style of vincent van gogh", "Vincent van Gogh"),
        ("a photo of tench", "tench"), ("artwork by the artist picasso in paint", "picasso"),
        ("painting by c0042", "[CLS]"), ("painting by c0042", "[EOS]"), ("painting by c0042", ""),
        ("painting by c0042", " "), ("a photo of a photo", "photo"), ("o'keeffe painting", "o'keeffe"),
        ("style of claude-monet, famous", "claude-monet"), ("paint in the style of x", "x"),
        ("an image of new york", "New York"), ("painting by zzz", "absent"),
    ]
    rows = []
    prompts = [c[0] for c in cases]
    enc = tok(prompts, return_tensors="pt", padding=True, truncation=True)
    for (p, s), ids in zip(cases, enc["input_ids"]):
        try:
            r = find_token_range(tok, ids, s)
            rows.append({"prompt": p, "subject": s, "ids": ids.tolist(), "range": [int(r[0]), int(r[1])]})
        except ValueError:
            rows.append({"prompt": p, "subject": s, "ids": ids.tolist(), "range": "ValueError"})
    with open(OUT / f"{tag}.json", "w") as f:
        json.dump(rows, f, indent=1)
    print(f"[golden] {tag}: {len(rows)} cases")


def main():
    scratch = Path(tempfile.mkdtemp(prefix="emcid_golden_"))
    try:
        em, ls, cz, HP, XLHP, ftr = import_reference(scratch)
        torch.set_num_threads(8)
        if "--only-uce" in sys.argv:
            golden_uce(scratch)
            return
        if "--only" in sys.argv:       # the heavy real-dimension summaries, one at a time (minutes each on 8 cores)
            which = sys.argv[sys.argv.index("--only") + 1]
            if which == "real_sd_n1000_summary":
                golden_sd(em, HP, scratch, which, "sd-v1.4", n_req=1000, layers=(7, 8, 9, 10), lam=4000, ew=0.5,
                          ragged=False, full=False, syllables=True, store_vstar=False)
            elif which == "real_sd_n1500_summary":      # the reference's largest shipped list size (erased-1500artists): Np = 1536
                golden_sd(em, HP, scratch, which, "sd-v1.4", n_req=1500, layers=(7, 8, 9, 10), lam=4000, ew=0.5,
                          ragged=False, full=False, syllables=True, store_vstar=False)
            elif which == "real_sd_outliers_summary":     # trained-weight-like statistics (syn.add_trained_like_outliers), N = 100
                golden_sd(em, HP, scratch, which, "sd-v1.4", n_req=100, layers=(7, 8, 9, 10), lam=4000, ew=0.5,
                          ragged=False, full=False, syllables=True, store_vstar=False, outliers=True)
            elif which == "real_sd_artist_n1000_summary":  # the shape of the reference's only 1 000-concept list: two-word artist names
                golden_sd(em, HP, scratch, which, "sd-v1.4", n_req=1000, layers=(7, 8, 9, 10), lam=4000, ew=0.5,
                          ragged=False, full=False, syllables="wide", store_vstar=False, names="artist")
            elif which == "real_sd_own_prompts_summary":   # no shared prefixes: every request's own three prompts, N = 100
                golden_sd(em, HP, scratch, which, "sd-v1.4", n_req=100, layers=(7, 8, 9, 10), lam=4000, ew=0.5,
                          ragged=False, full=False, syllables=True, store_vstar=False, own_prompts=True)
            elif which == "real_sd_own_prompts_n1000_summary":   # ~36 000 trie rows: the projections' 160 x 128 tile form end to end
                golden_sd(em, HP, scratch, which, "sd-v1.4", n_req=1000, layers=(7, 8, 9, 10), lam=4000, ew=0.5,
                          ragged=False, full=False, syllables=True, store_vstar=False, own_prompts=True)
            elif which == "real_sdxl_summary":
                golden_sdxl_real(em, XLHP, scratch)
            elif which == "real_sdxl_n1000_summary":     # BASELINE config 4 at its full size (N = 1000: the solver's
                golden_sdxl_real(em, XLHP, scratch, tag=which, n_req=1000)      # two-round stream-K / pairs path at d = 5120)
            elif which == "real_stage0_summary":
                golden_stage0_real(ls, scratch)
            elif which == "toy_extras":
                golden_toy_extras(cz, ls, scratch)
            elif which == "toy_stage1":
                golden_stage1(cz, HP, scratch)
            elif which == "toy_stage1_sdxl":
                golden_stage1_sdxl(cz, XLHP, scratch)
            elif which == "toy_stage1_more":
                golden_stage1_more(cz, HP, scratch)
            elif which == "toy_stage1_global":
                golden_stage1_global(cz, HP, scratch)
            elif which == "toy_stage1_v1":
                golden_stage1_v1(cz, HP, scratch)
            elif which == "toy_xattn_stage1":
                golden_xattn_stage1(cz, HP, scratch)
            elif which == "toy_multi_token":
                golden_multi_token(em, HP, scratch)
            else:
                raise SystemExit(f"unknown fixture {which}")
            return
        golden_token_ranges(ftr)
        golden_uce(scratch)
        golden_sd(em, HP, scratch, "toy_sd", "toy", n_req=8, layers=(1, 2, 3, 4), lam=50, ew=0.6, ragged=True, full=True)
        golden_sdxl(em, XLHP, scratch)
        golden_stage0(ls, scratch)
        golden_xattn(em, HP, scratch)
        golden_cal_insert(em, HP, scratch)
        golden_instruction(em, HP, scratch)
        golden_toy_extras(cz, ls, scratch)
        golden_stage1(cz, HP, scratch)
        golden_stage1_sdxl(cz, XLHP, scratch)
        golden_stage1_more(cz, HP, scratch)
        golden_stage1_global(cz, HP, scratch)
        golden_stage1_v1(cz, HP, scratch)
        golden_xattn_stage1(cz, HP, scratch)
        golden_multi_token(em, HP, scratch)
        if "--skip-real" not in sys.argv:
            golden_sd(em, HP, scratch, "real_sd_summary", "sd-v1.4", n_req=24, layers=(7, 8, 9, 10), lam=4000,
                      ew=0.5, ragged=False, full=False)
    finally:
        os.chdir(REPO)
        shutil.rmtree(scratch, ignore_errors=True)
    n_ref = sum(1 for _ in REF.rglob("*") if _.is_file())
    print(f"[golden] reference tree files: {n_ref} (must stay 111; no __pycache__ written)")


if __name__ == "__main__":
    main()
