"""ORACLE — CPU restatement of EMCID's closed-form mass-edit path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg may import it.  The product (emcid_amd/) never does.

It restates, op-for-op in PyTorch-CPU / numpy, what the reference computes on this
path (two full hooked encoder forwards per edited layer, per-prompt Python gathers,
fp32 residuals, fp64 `torch.linalg.solve`, fp32 `a.t().mm(a)` second moment), each
function citing the reference file:line it follows.  Third-party arithmetic at the
boundary is the same as the reference's: HF `CLIPTextModel.forward`
(reference pins transformers==4.27.4, environment.yaml:31; this image has 5.15) and
`torch.linalg.solve` (LAPACK getrf/getrs; reference pins torch==2.0.1).

PARITY PIN: the reference holds no golden vectors for this path, so the
oracle is pinned against outputs of the reference itself, run in the build container by
tests/golden/make_golden.py and committed as tests/golden/*.npz (tests/test_oracle_golden.py).
"""
from __future__ import annotations

import copy
import json
import math
import random
import unicodedata
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

# --------------------------------------------------------------------------------------
# module lookup (reference: util/nethook.py:375-392)
# --------------------------------------------------------------------------------------

def _resolve(model, dotted: str, table) -> object:
    """Named lookup; tolerates the transformers-4.x `text_model.` prefix on a 5.x model
    and vice versa (SURVEY.md §8c shim 3 — data-level rename, no arithmetic)."""
    names = dict(table)
    if dotted in names:
        return names[dotted]
    alt = dotted[len("text_model."):] if dotted.startswith("text_model.") else "text_model." + dotted
    if alt in names:
        return names[alt]
    raise LookupError(dotted)


def get_module(model, name):
    return _resolve(model, name, model.named_modules())


def get_parameter(model, name):
    return _resolve(model, name, model.named_parameters())


# --------------------------------------------------------------------------------------
# token range (reference: experiments/causal_trace.py:1046-1103)
# --------------------------------------------------------------------------------------

def find_token_range(tokenizer, token_array, substring_orig: str) -> Tuple[int, int]:
    sub = substring_orig
    n = len(token_array)
    if sub == "[CLS]":
        return (0, 1)
    if sub in ("[EOS]", "", " "):
        return (n - 1, n)
    sub = sub.replace(" ", "").lower()
    pieces = [tokenizer.decode([t]) for t in token_array]
    whole = tokenizer.decode(token_array).replace(" ", "")
    if "’" in sub:
        whole = whole.replace("'", "’")
    whole = unicodedata.normalize("NFKC", whole)
    sub = unicodedata.normalize("NFKC", sub)
    start_char = whole.index(sub)  # ValueError when absent, like the reference
    end_char = start_char + len(sub)
    pos = 0
    first = last = None
    for i, piece in enumerate(pieces):
        if not ("ń" in sub and int(token_array[i]) == 78):
            pos += len(piece)
        if first is None and pos > start_char:
            first = i
        if last is None and pos >= end_char:
            last = i + 1
            break
    return (first, last)


# --------------------------------------------------------------------------------------
# K/Z assembly (reference: emcid/compute_z.py:56-74, 2252-2325; emcid/compute_ks.py:21-41)
# --------------------------------------------------------------------------------------

def tokenize_prompts(prompts: List[str], tokenizer, device, padding_length=None):
    """compute_z.py:56-74."""
    if padding_length is None:
        enc = tokenizer(prompts, return_tensors="pt", padding=True, truncation=True)
    else:
        enc = tokenizer(prompts, return_tensors="pt", padding="max_length", truncation=True, max_length=padding_length)
    return {k: v.to(device) for k, v in enc.items()}


def expand_requests(requests: Sequence[Dict]) -> Tuple[List[str], List[str], List[int]]:
    """Prompt strings, per-prompt subject, prompts-per-request (compute_z.py:2270-2283, 2318-2320)."""
    prompts, subjects, counts = [], [], []
    pre = "source_prompts" in requests[0]
    for r in requests:
        ps = list(r["source_prompts"]) if pre else [p.format(r["source"]) for p in r["prompts"]]
        prompts += ps
        subjects += [r["source"]] * len(ps)
        counts.append(len(r["prompts"]) if "prompts" in requests[0] else len(r["source_prompts"]))
    return prompts, subjects, counts


def module_input_output_at_words_multi(text_encoder, tokenizer, requests, module_name, num_fact_token):
    """The num_fact_token > 1 branch (compute_z.py:2329-2382): prompts padded to (longest + num_fact_token - 2) tokens
    with padding="max_length"; per prompt the rows [last subject token, EOS, the num_fact_token - 2 positions after it];
    returns (N, num_fact_token, d) and (N, num_fact_token, h) means over each request's prompts."""
    device = next(text_encoder.parameters()).device
    prompts, subjects, counts = expand_requests(requests)
    first = tokenize_prompts(prompts, tokenizer, device)
    n_pad = num_fact_token - 2
    enc = tokenizer(prompts, return_tensors="pt", padding="max_length", truncation=True,
                    max_length=len(first["input_ids"][0]) + n_pad)
    inp = {k: v.to(device) for k, v in enc.items()}
    lookup = [[find_token_range(tokenizer, ids, w)[-1] - 1] for ids, w in zip(inp["input_ids"], subjects)]
    eos = [int(m.sum()) - 1 for m in inp["attention_mask"]]
    lookup = [lk + list(range(e, e + n_pad + 1)) for lk, e in zip(lookup, eos)]
    cap = {}

    def hook(mod, args, out):
        cap["in"], cap["out"] = args[0], out

    h = get_module(text_encoder, module_name).register_forward_hook(hook)
    try:
        with torch.no_grad():
            text_encoder(**inp)
    finally:
        h.remove()
    rows_in = torch.stack([cap["in"][i, idx, :] for i, idx in enumerate(lookup)], 0).detach().clone()
    rows_out = torch.stack([cap["out"][i, idx, :] for i, idx in enumerate(lookup)], 0).detach().clone()
    edges = np.cumsum([0] + counts).tolist()
    k = torch.stack([rows_in[edges[i]:edges[i + 1]].mean(0) for i in range(len(requests))], 0)
    z = torch.stack([rows_out[edges[i]:edges[i + 1]].mean(0) for i in range(len(requests))], 0)
    return k, z


def module_input_output_at_words(text_encoder, tokenizer, requests, module_name) -> Tuple[torch.Tensor, torch.Tensor]:
    """One full encoder forward on all N*P prompts; fc2 input/output at the last subject
    token of each prompt; mean over each request's prompts (num_fact_token == 1 branch)."""
    device = next(text_encoder.parameters()).device
    prompts, subjects, counts = expand_requests(requests)
    inp = tokenize_prompts(prompts, tokenizer, device)
    lookup = [find_token_range(tokenizer, ids, w)[-1] - 1 for ids, w in zip(inp["input_ids"], subjects)]
    cap = {}

    def hook(mod, args, out):
        cap["in"], cap["out"] = args[0], out

    h = get_module(text_encoder, module_name).register_forward_hook(hook)
    try:
        with torch.no_grad():
            text_encoder(**inp)
    finally:
        h.remove()
    rows_in = torch.stack([cap["in"][i, j, :] for i, j in enumerate(lookup)], 0).detach().clone()
    rows_out = torch.stack([cap["out"][i, j, :] for i, j in enumerate(lookup)], 0).detach().clone()
    edges = np.cumsum([0] + counts).tolist()
    k = torch.stack([rows_in[edges[i]:edges[i + 1]].mean(0) for i in range(len(requests))], 0)
    z = torch.stack([rows_out[edges[i]:edges[i + 1]].mean(0) for i in range(len(requests))], 0)
    return k, z


# --------------------------------------------------------------------------------------
# Stage 1 (reference: emcid/compute_z.py:34-53 preprocess_img, :315-649 compute_z_text_encoder)
# --------------------------------------------------------------------------------------

def preprocess_img(images, resolution: int) -> torch.Tensor:
    """Resize(bilinear) -> CenterCrop -> RandomHorizontalFlip -> ToTensor -> Normalize(0.5, 0.5) on PIL images
    (compute_z.py:34-53; torchvision's transforms restated with PIL + torch: one ``torch.rand(1)`` per image for the flip)."""
    from PIL import Image
    out = []
    for im in images:
        im = im.convert("RGB")
        w, h = im.size
        if min(w, h) != resolution:                       # Resize(int): the shorter side becomes `resolution`
            if w <= h:
                im = im.resize((resolution, int(resolution * h / w)), Image.BILINEAR)
            else:
                im = im.resize((int(resolution * w / h), resolution), Image.BILINEAR)
            w, h = im.size
        left, top = int(round((w - resolution) / 2.0)), int(round((h - resolution) / 2.0))
        im = im.crop((left, top, left + resolution, top + resolution))
        if torch.rand(1) < 0.5:
            im = im.transpose(Image.FLIP_LEFT_RIGHT)
        x = torch.from_numpy(np.asarray(im, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0)
        out.append((x - 0.5) / 0.5)
    return torch.stack(out)


def _hidden(out):
    return out[0] if isinstance(out, tuple) else out      # transformers 4.x layers return a tuple, 5.x the tensor


def compute_z_sdxl_text_encoders(pipe, request: Dict, hparams: Dict, layers, resolution: int = 512):
    """Stage 1 of the SDXL pair (compute_z.py:651-1037), op for op: ONE Adam over (delta, deltas_2) added at each prompt's last
    subject token in ``layer_module_tmp.format(layer)`` of text_encoder / text_encoder_2 (hooks on the pipeline's own encoders,
    active only around the edited forwards); per step one VAE encode, the clean destination forwards of both encoders, the
    edited source forwards, two UNet forwards with the concatenated penultimate hidden states as context and the second
    encoder's text_embeds + size ids as added conditions; MSE + both weight decays (+ both pooled-output alignment terms);
    both deltas projected onto their L2 balls.  Reference quirks kept: the destination forward of the SECOND encoder is fed
    the FIRST tokenizer's ids (:842-843); ``noise_scheduler = pipe.scheduler`` (:741).  Returns (v*, v*_2)."""
    import torch.nn.functional as F
    hp = lambda k, d=None: hparams.get(k, d)
    device = next(pipe.text_encoder.parameters()).device
    layer, layer_2 = layers
    te1, te2 = pipe.text_encoder, pipe.text_encoder_2
    source_prompts = [p.format(request["source"]) for p in request["prompts"]]
    objective = hp("objective")
    dest_prompts = ["" for _ in request["prompts"]] if objective == "esd" else [p.format(request["dest"]) for p in request["prompts"]]
    delta = torch.zeros((te1.config.hidden_size,), requires_grad=True, device=device)
    deltas_2 = torch.zeros((te2.config.hidden_size,), requires_grad=True, device=device)
    state = {"source_init": None, "source_init_2": None, "edit": False}
    opt = torch.optim.Adam([delta, deltas_2], lr=hp("v_lr"))
    noise_scheduler = pipe.scheduler
    for m in (pipe.vae, pipe.unet, te1, te2):
        for prm in m.parameters():
            prm.requires_grad = False
    spp = hp("samples_per_prompt", 1)
    if objective not in ("ablate-source", "ablate-dest"):
        raise ValueError(f"Objective {objective} can not be used for compute_z.")
    if "training_img_paths" in request:
        from PIL import Image
        all_imgs = [Image.open(path) for path in request["training_img_paths"]]
    elif "images" in request:
        all_imgs = request["images"]
    else:
        generator = torch.Generator(device).manual_seed(int(request["seed_train"])) if request["seed_train"] is not None else None
        all_imgs = []
        with torch.no_grad():
            if objective == "ablate-source":
                for _ in range(spp):
                    all_imgs.extend(pipe(source_prompts, guidance_scale=7.5, generator=generator).images)
            else:
                for _ in range(spp):
                    for prompt in source_prompts:
                        all_imgs.append(pipe(prompt, guidance_scale=7.5, generator=generator).images[0])
    all_imgs = preprocess_img(all_imgs, resolution)
    bsz = len(source_prompts)
    all_imgs = all_imgs.reshape(spp, bsz, *all_imgs.shape[1:]).transpose(0, 1)       # "(s b) c h w -> b s c h w"
    assert len(all_imgs) % bsz == 0
    src_inp = tokenize_prompts(source_prompts, pipe.tokenizer, device)
    dst_inp = tokenize_prompts(dest_prompts, pipe.tokenizer, device)
    src_inp_2 = tokenize_prompts(source_prompts, pipe.tokenizer_2, device)
    dst_inp_2 = tokenize_prompts(dest_prompts, pipe.tokenizer_2, device)
    src_lookup = [find_token_range(pipe.tokenizer, ids, request["source"])[-1] - 1 for ids in src_inp["input_ids"]]
    src_lookup_2 = [find_token_range(pipe.tokenizer_2, ids, request["source"])[-1] - 1 for ids in src_inp_2["input_ids"]]
    [find_token_range(pipe.tokenizer, ids, request["dest"]) for ids in dst_inp["input_ids"]]          # (:818-821, :827-829: computed,
    [find_token_range(pipe.tokenizer_2, ids, request["dest"]) for ids in dst_inp_2["input_ids"]]      #  raise if the dest is absent)
    assert len(src_inp["input_ids"]) == len(dst_inp["input_ids"]) == len(all_imgs)

    def make_hook(which, lookup, dvec):
        def hook(mod, args, out):
            if not state["edit"]:
                return out
            h = _hidden(out)
            if state[which] is None:
                state[which] = h[0, lookup[0]].detach().clone()
            for i, idx in enumerate(lookup):
                if hp("replace_repr", False):
                    h[i, idx, :] = dvec
                else:
                    h[i, idx, :] += dvec
            return out
        return hook

    handles = [get_module(te1, hparams["layer_module_tmp"].format(layer)).register_forward_hook(make_hook("source_init", src_lookup, delta)),
               get_module(te2, hparams["layer_module_tmp"].format(layer_2)).register_forward_hook(make_hook("source_init_2", src_lookup_2, deltas_2))]
    try:
        for it in range(hp("v_num_grad_steps")):
            opt.zero_grad()
            sample_indices = torch.randint(0, spp, (bsz,))
            img_batch = all_imgs[torch.arange(bsz), sample_indices].to(device)
            with torch.no_grad():
                latents = pipe.vae.encode(img_batch).latent_dist.sample() * pipe.vae.config.scaling_factor
                d1 = te1(**dst_inp, output_hidden_states=True)
                dest_txt, dest_pool = d1.hidden_states[-2], d1.pooler_output
                d2 = te2(**dst_inp, output_hidden_states=True)                 # the FIRST tokenizer's ids, as the reference (:842)
                dest_txt_2, dest_pool_2 = d2.hidden_states[-2], d2.text_embeds
                dest_embeds = torch.cat([dest_txt, dest_txt_2], dim=-1)
                height = width = pipe.default_sample_size * pipe.vae_scale_factor
                add_time_ids = pipe._get_add_time_ids(original_size=(height, width), crops_coords_top_left=(0, 0),
                                                      source_size=(height, width), dtype=dest_embeds.dtype,
                                                      text_encoder_projection_dim=te2.config.projection_dim)
                add_time_ids = add_time_ids.repeat(bsz, 1).to(device)
                dest_cond = {"text_embeds": dest_pool_2, "time_ids": add_time_ids}
            noise = torch.randn_like(latents, device=device)
            timesteps = torch.randint(0, noise_scheduler.config.num_train_timesteps, (bsz,), device=device).long()
            noisy = noise_scheduler.add_noise(latents, noise, timesteps)
            state["edit"] = True
            try:
                e1 = te1(**src_inp, output_hidden_states=True)
                e2 = te2(**src_inp_2, output_hidden_states=True)
            finally:
                state["edit"] = False
            edit_txt, edit_pool = e1.hidden_states[-2], e1.pooler_output
            edit_txt_2, edit_pool_2 = e2.hidden_states[-2], e2.text_embeds
            edit_embeds = torch.cat([edit_txt, edit_txt_2], dim=-1)
            edit_cond = {"text_embeds": edit_pool_2, "time_ids": add_time_ids}
            if not hp("no_noise_loss", False):
                edit_pred = pipe.unet(noisy, timesteps, encoder_hidden_states=edit_embeds, added_cond_kwargs=edit_cond).sample
                pred_dest = pipe.unet(noisy, timesteps, encoder_hidden_states=dest_embeds, added_cond_kwargs=dest_cond).sample
            if hp("use_sampled_noise", False) or request.get("use_real_noise", False):
                mse = F.mse_loss(noise, edit_pred, reduction="mean")
            elif hp("no_noise_loss", False):
                mse = None
            else:
                mse = F.mse_loss(edit_pred, pred_dest, reduction="mean")
            reg = hp("v_weight_decay") * (torch.norm(delta) / torch.norm(state["source_init"]) ** 2)
            reg_2 = hp("v_weight_decay") * (torch.norm(deltas_2) / torch.norm(state["source_init_2"]) ** 2)
            loss = reg + reg_2 if mse is None else mse + reg + reg_2
            if hp("cal_text_repr_loss", False) and request.get("txt_align", True):
                scale = hp("text_repr_loss_scale_factor")
                loss = loss + scale * F.mse_loss(edit_pool, dest_pool, reduction="mean") \
                    + scale * F.mse_loss(edit_pool_2, dest_pool_2, reduction="mean")
            loss.backward()
            opt.step()
            max_norm = hp("clamp_norm_factor") * state["source_init"].norm()
            max_norm_2 = hp("clamp_norm_factor") * state["source_init_2"].norm()
            if delta.norm() > max_norm:
                with torch.no_grad():
                    delta[...] = delta * max_norm / delta.norm()
            if deltas_2.norm() > max_norm_2:
                with torch.no_grad():
                    deltas_2[...] = deltas_2 * max_norm_2 / deltas_2.norm()
    finally:
        for h_ in handles:
            h_.remove()
    return (state["source_init"] + delta).detach(), (state["source_init_2"] + deltas_2).detach()


# the path the reference reads its Fisher statistics from, relative to the cwd (compute_z.py:482); tests point it at a fixture
FIM_FILE = "data/fim_stats/text_encoder/ccs_filtered_stats/text_model.encoder.layers.10.mlp.fc2_float32_mean_step10_3000.npz"


def compute_z_text_encoder(pipe, request: Dict, hparams: Dict, layer: int, noise_scheduler, resolution: int = 512) -> torch.Tensor:
    """Per-concept Adam optimisation of v* through the UNet (compute_z.py:315-649), op for op: deep copy of the encoder
    with a hook that adds ``delta`` at each prompt's last subject token in ``layer_module_tmp.format(layer)``'s output,
    per step one VAE encode, the clean text-encoder forwards, two (esd: three) UNet forwards, MSE + weight decay (+ the
    text-alignment term), Adam step, projection onto the L2 ball.  ``hparams``: the JSON fields (missing flags = the
    dataclass defaults of emcid_hparams.py:55-163).  Randomness: the global torch generators, consumed in the
    reference's order."""
    from copy import deepcopy
    import torch.nn.functional as F
    hp = lambda k, d=None: hparams.get(k, d)
    device = next(pipe.text_encoder.parameters()).device
    fim = None
    if hp("use_ewc", False):
        # compute_z.py:478-486: the Fisher diagonal as a CombinedStat(mean=Mean()) npz at a cwd-relative path (FIM_FILE here)
        with np.load(FIM_FILE, allow_pickle=True) as data:
            fim = torch.from_numpy(np.asarray(data["mean.mean"])).to(device)
    te_edit = deepcopy(pipe.text_encoder).to(device)
    source_prompts = [p.format(request["source"]) for p in request["prompts"]]
    objective = hp("objective")
    dest_prompts = ["" for _ in request["prompts"]] if objective == "esd" else [p.format(request["dest"]) for p in request["prompts"]]
    delta = torch.zeros((te_edit.config.hidden_size,), requires_grad=True, device=device)
    state = {"source_init": None}
    opt = torch.optim.Adam([delta], lr=hp("v_lr"))
    for m in (te_edit, pipe.vae, pipe.unet, pipe.text_encoder):
        for prm in m.parameters():
            prm.requires_grad = False
    spp = hp("samples_per_prompt", 1)
    if objective not in ("ablate-source", "ablate-dest", "esd"):
        raise ValueError(f"Objective {objective} can not be used for compute_z.")
    if "training_img_paths" in request and objective != "esd":
        from PIL import Image
        all_imgs = [Image.open(path) for path in request["training_img_paths"]]
    elif "images" in request and objective != "esd":
        all_imgs = request["images"]
    else:
        generator = torch.Generator(device).manual_seed(int(request["seed_train"])) if request["seed_train"] is not None else None
        all_imgs = []
        for _ in range(spp):
            all_imgs.extend(pipe(source_prompts, guidance_scale=7.5, generator=generator).images)
    all_imgs = preprocess_img(all_imgs, resolution)
    bsz = len(source_prompts)
    all_imgs = all_imgs.reshape(spp, bsz, *all_imgs.shape[1:]).transpose(0, 1)       # "(s b) c h w -> b s c h w"
    assert len(all_imgs) % bsz == 0
    src_inp = tokenize_prompts(source_prompts, pipe.tokenizer, device)
    dst_inp = tokenize_prompts(dest_prompts, pipe.tokenizer, device)
    tok = pipe.tokenizer
    if hp("align_obj_eos_pad", False):
        full = lambda ps: {k: v.to(device) for k, v in tok(ps, max_length=tok.model_max_length, return_tensors="pt",
                                                           padding="max_length", truncation=True).items()}
        src_full, dst_full = full(source_prompts), full(dest_prompts)
        src_eos = [int(m.sum()) - 1 for m in src_full["attention_mask"]]
        dst_eos = [int(m.sum()) - 1 for m in dst_full["attention_mask"]]
        far = max(src_eos + dst_eos)
        src_slices = [list(range(e, tok.model_max_length - max(0, far - e))) for e in src_eos]
        dst_slices = [list(range(e, tok.model_max_length - max(0, far - e))) for e in dst_eos]
        with torch.no_grad():
            dest_full = pipe.text_encoder(**dst_full)[0]
    if hp("contrastive_text_loss", False):
        neg_inp = tokenize_prompts(request["negative_prompts"], tok, device)
    src_lookup = [find_token_range(tok, ids, request["source"])[-1] - 1 for ids in src_inp["input_ids"]]
    dst_lookup = [find_token_range(tok, ids, request["dest"])[-1] - 1 for ids in dst_inp["input_ids"]]
    assert len(src_inp["input_ids"]) == len(dst_inp["input_ids"]) == len(all_imgs)
    layer_mod = get_module(te_edit, hparams["layer_module_tmp"].format(layer))

    def hook(mod, args, out):
        h = _hidden(out)
        if state["source_init"] is None:
            state["source_init"] = h[0, src_lookup[0]].detach().clone()
        for i, idx in enumerate(src_lookup):
            if hp("replace_repr", False):
                h[i, idx, :] = delta
            else:
                h[i, idx, :] += delta
        return out

    handle = layer_mod.register_forward_hook(hook)
    try:
        for it in range(hp("v_num_grad_steps")):
            opt.zero_grad()
            sample_indices = torch.randint(0, spp, (bsz,))
            img_batch = all_imgs[torch.arange(bsz), sample_indices].to(device)
            with torch.no_grad():
                latents = pipe.vae.encode(img_batch).latent_dist.sample() * pipe.vae.config.scaling_factor
                dest_repr, dest_pool = pipe.text_encoder(**dst_inp)[0:2]
                if objective == "esd" or hp("cal_text_repr_loss", False):
                    source_repr = pipe.text_encoder(**src_inp)[0]
                    if hp("contrastive_text_loss", False):
                        neg_repr, neg_pool = pipe.text_encoder(**neg_inp)[0:2]
            noise = torch.randn_like(latents, device=device)
            timesteps = torch.randint(0, noise_scheduler.config.num_train_timesteps, (bsz,), device=device).long()
            noisy = noise_scheduler.add_noise(latents, noise, timesteps)
            edit_repr, edit_pool = te_edit(**src_inp)[0:2]
            with torch.no_grad():
                if objective == "esd":
                    pred_source = pipe.unet(noisy, timesteps, source_repr).sample
            if not hp("no_noise_loss", False):
                edit_pred = pipe.unet(noisy, timesteps, edit_repr).sample
                pred_dest = pipe.unet(noisy, timesteps, dest_repr).sample
            source_init = state["source_init"]
            if fim is not None and "ablate" in objective:       # :547-549 (the esd branch keeps the weight decay, :553)
                reg = torch.sum(float(hp("ewc_lambda", 1e4)) * fim * delta ** 2) / (2 * torch.norm(source_init) ** 2)
            else:
                reg = hp("v_weight_decay") * (torch.norm(delta) / torch.norm(source_init) ** 2)
            if "ablate" in objective:
                if hp("use_sampled_noise", False) or request.get("use_real_noise", False):
                    mse = F.mse_loss(noise, edit_pred, reduction="mean")
                elif hp("no_noise_loss", False):
                    mse = None
                else:
                    mse = F.mse_loss(edit_pred, pred_dest, reduction="mean")
                loss = reg if hp("no_noise_loss", False) else mse + reg
            else:
                tmp = pred_dest - hp("esd_mu") * (pred_source - pred_dest)
                loss = F.mse_loss(edit_pred, tmp, reduction="mean") + reg
            if hp("cal_text_repr_loss", False) and request.get("txt_align", True):
                scale = hp("text_repr_loss_scale_factor")
                ar = torch.arange(bsz)
                if hp("contrastive_text_loss", False):
                    single = pipe.text_encoder(**tokenize_prompts([request["dest"]], tok, device))[1]
                    emb = torch.cat([single, neg_pool], dim=0)
                    scores = torch.squeeze(-torch.cdist(edit_pool.unsqueeze(0), emb.unsqueeze(0)))
                    loss = loss + scale * (-torch.log_softmax(scores, dim=1)[:, 0].mean(dim=0))
                elif hp("align_object_token", False):
                    loss = loss + scale * F.mse_loss(edit_repr[ar, src_lookup, :], dest_repr[ar, dst_lookup, :], reduction="mean")
                elif hp("align_obj_eos_pad", False):
                    e_full = te_edit(**src_full)[0]
                    e_pad = torch.stack([e_full[i, sl, :] for i, sl in enumerate(src_slices)], dim=0)
                    d_pad = torch.stack([dest_full[i, sl, :] for i, sl in enumerate(dst_slices)], dim=0)
                    loss = loss + scale * F.mse_loss(torch.cat([edit_repr[ar, src_lookup, :].unsqueeze(1), e_pad], dim=1),
                                                     torch.cat([dest_repr[ar, dst_lookup, :].unsqueeze(1), d_pad], dim=1),
                                                     reduction="mean")
                else:
                    loss = loss + scale * F.mse_loss(edit_pool, dest_pool, reduction="mean")
            loss.backward()
            opt.step()
            max_norm = hp("clamp_norm_factor") * source_init.norm()
            if delta.norm() > max_norm:
                with torch.no_grad():
                    delta[...] = delta * max_norm / delta.norm()
    finally:
        handle.remove()
    return (state["source_init"] + delta).detach()


SLD_PRESETS = {     # compute_z.py:174-191 (the "max" / "strong" configurations of safe latent diffusion)
    "max": dict(sld_guidance_scale=5000, sld_warmup_steps=0, sld_threshold=1.0, sld_momentum_scale=0.5, sld_mom_beta=0.7),
    "strong": dict(sld_guidance_scale=2000, sld_warmup_steps=7, sld_threshold=0.025, sld_momentum_scale=0.5, sld_mom_beta=0.7),
}


def compute_z_text_encoder_global(pipe, request: Dict, hparams: Dict, layer: int, noise_scheduler, resolution: int = 512) -> torch.Tensor:
    """The ``sld_supervision`` Stage 1 of a global concept (compute_z.py:77-312; selected at emcid_main.py:911-918), op for op: a
    deep copy of the encoder whose ``layer_module_tmp.format(layer)`` output gets ``delta`` added at position 0 ("[CLS]") or -1
    ("[EOS]") of EVERY source prompt; the latents are sampled ONCE before the loop (:203-205); per step latent noise, timesteps,
    the edited encoder forward, the clean UNet predictions under the source / unconditional / safety embeddings, the
    safe-latent-diffusion guidance (:232-248), the edited UNet prediction, MSE + weight decay, Adam, the L2 ball.  Training
    images: ``training_img_paths`` / ``images``, or one per prompt from the pipeline with ``request["seeds"]`` (:140-146, :159-164);
    the reference's own sampler of ablate-dest images (``sld_generate``: a hub pipeline, :155) is not restated."""
    from copy import deepcopy
    import torch.nn.functional as F
    hp = lambda k, d=None: hparams.get(k, d)
    device = next(pipe.text_encoder.parameters()).device
    te_edit = deepcopy(pipe.text_encoder).to(device)
    source_prompts = request["source_prompts"]
    delta = torch.zeros((te_edit.config.hidden_size,), requires_grad=True, device=device)
    state = {"source_init": None}
    if request["source"] == "[CLS]":
        edit_idx = 0
    elif request["source"] == "[EOS]":
        edit_idx = -1
    else:
        raise NameError("name 'edit_idx' is not defined")       # the reference's hook stops there (:108-111, :123)
    opt = torch.optim.Adam([delta], lr=hp("v_lr"))
    for m in (te_edit, pipe.vae, pipe.unet, pipe.text_encoder):
        for prm in m.parameters():
            prm.requires_grad = False
    objective = hp("objective")
    if objective not in ("ablate-source", "ablate-dest", "esd"):
        raise ValueError(f"Objective {objective} can not be used for compute_z.")
    if objective != "esd" and "training_img_paths" in request:
        from PIL import Image
        imgs = [Image.open(path) for path in request["training_img_paths"]]
    elif objective != "esd" and "images" in request:
        imgs = request["images"]
    elif objective == "ablate-dest":
        raise NotImplementedError("sld_generate (compute_z.py:155) is not restated: pass the training images")
    else:
        imgs = []
        for prompt, seed in zip(source_prompts, request["seeds"]):
            generator = torch.Generator(device).manual_seed(int(seed)) if seed is not None else None
            imgs.append(pipe([prompt], guidance_scale=7.5, generator=generator).images[0])
    img_batch = preprocess_img(imgs, resolution).to(device)
    src_inp = tokenize_prompts(source_prompts, pipe.tokenizer, device)
    safe_inp = tokenize_prompts(request["safe_words"], pipe.tokenizer, device)
    uncond_inp = tokenize_prompts([""] * img_batch.shape[0], pipe.tokenizer, device)
    assert len(src_inp["input_ids"]) == len(img_batch), "The number of prompts and images should be the same."
    bsz = len(img_batch)
    if hp("sld_type", "max") not in SLD_PRESETS:
        raise ValueError(f"sld_type {hp('sld_type')} not supported")
    sld = {k: torch.tensor(v).to(device) for k, v in SLD_PRESETS[hp("sld_type", "max")].items()}
    with torch.no_grad():
        latents = pipe.vae.encode(img_batch).latent_dist.sample() * pipe.vae.config.scaling_factor
        safety_repr = pipe.text_encoder(**safe_inp)[0]
        source_repr = pipe.text_encoder(**src_inp)[0]
        uncond_repr = pipe.text_encoder(**uncond_inp)[0]

    def hook(mod, args, out):
        h = _hidden(out)
        if state["source_init"] is None:
            state["source_init"] = h[:, edit_idx].detach().clone().mean(dim=0)
        for i in range(bsz):
            h[i, edit_idx, :] += delta
        return out

    handle = get_module(te_edit, hparams["layer_module_tmp"].format(layer)).register_forward_hook(hook)
    try:
        for it in range(hp("v_num_grad_steps")):
            opt.zero_grad()
            noise = torch.randn_like(latents, device=device)
            timesteps = torch.randint(0, noise_scheduler.config.num_train_timesteps, (bsz,), device=device).long()
            noisy = noise_scheduler.add_noise(latents, noise, timesteps)
            edit_repr = te_edit(**src_inp)[0]
            with torch.no_grad():
                pred_source = pipe.unet(noisy, timesteps, source_repr).sample
                pred_uncond = pipe.unet(noisy, timesteps, uncond_repr).sample
                if hp("sld_supervision", False):
                    pred_safety = pipe.unet(noisy, timesteps, safety_repr).sample
                    scale = torch.clamp(torch.abs((pred_source - pred_safety)) * sld["sld_guidance_scale"], max=1.0)
                    concept_scale = torch.where((pred_source - pred_safety) >= sld["sld_threshold"], torch.zeros_like(scale), scale)
                    guidance = torch.mul((pred_safety - pred_uncond), concept_scale)
            edit_pred = pipe.unet(noisy, timesteps, edit_repr).sample
            source_init = state["source_init"]
            if "ablate" in objective:
                if hp("use_sampled_noise", False):
                    mse = F.mse_loss(noise, edit_pred, reduction="mean")
                else:
                    mse = F.mse_loss(edit_pred, pred_source - guidance, reduction="mean")
                loss = mse + hp("v_weight_decay") * (torch.norm(delta) / torch.norm(source_init) ** 2)
            else:
                tmp = pred_uncond - hp("esd_mu") * (pred_source - pred_uncond)
                loss = F.mse_loss(edit_pred, tmp, reduction="mean") + hp("v_weight_decay") * (torch.norm(delta) / torch.norm(source_init) ** 2)
            loss.backward()
            opt.step()
            max_norm = hp("clamp_norm_factor") * source_init.norm()
            if delta.norm() > max_norm:
                with torch.no_grad():
                    delta[...] = delta * max_norm / delta.norm()
    finally:
        handle.remove()
    return (state["source_init"] + delta).detach()


def compute_z_text_encoder_v1(pipe, request: Dict, hparams: Dict, layer: int, noise_scheduler, clip_towers, resolution: int = 512) -> torch.Tensor:
    """The ``txt_img_align_scale_factor != 0`` Stage 1 (compute_z.py:1360-1648; selected at emcid_main.py:919-926), op for op.
    ``clip_towers`` = (CLIPTextModelWithProjection, CLIPVisionModelWithProjection, CLIPProcessor): what the reference loads from
    the hub (openai/clip-vit-large-patch14, :1376-1378, :1440).  The hooked model is the text tower with projection; the latents are
    sampled once before the loop (:1483-1485); text-alignment terms in the projected space; the image-alignment term against the
    CLIP embedding of the training images (ablate-dest)."""
    import torch.nn.functional as F
    hp = lambda k, d=None: hparams.get(k, d)
    device = next(pipe.text_encoder.parameters()).device
    te_edit, vision, processor = clip_towers
    te_edit = te_edit.to(device)
    tok = pipe.tokenizer
    source_prompts = [p.format(request["source"]) for p in request["prompts"]]
    objective = hp("objective")
    dest_prompts = ["" for _ in request["prompts"]] if objective == "esd" else [p.format(request["dest"]) for p in request["prompts"]]
    delta = torch.zeros((te_edit.config.hidden_size,), requires_grad=True, device=device)
    state = {"source_init": None}
    opt = torch.optim.Adam([delta], lr=hp("v_lr"))
    for m in (te_edit, pipe.vae, pipe.unet, pipe.text_encoder):
        for prm in m.parameters():
            prm.requires_grad = False
    generator = torch.Generator(device).manual_seed(int(request["seed_train"])) if request["seed_train"] is not None else None
    if objective not in ("ablate-source", "ablate-dest", "esd"):
        raise ValueError(f"Objective {objective} can not be used for compute_z.")
    if objective != "esd" and "training_img_paths" in request:
        from PIL import Image
        imgs = [Image.open(path) for path in request["training_img_paths"]]
    elif objective != "esd" and "images" in request:
        imgs = request["images"]
    else:
        imgs = pipe(dest_prompts if objective == "ablate-dest" else source_prompts, guidance_scale=7.5, generator=generator).images
    if objective == "ablate-dest" and request["txt_img_align"]:
        with torch.no_grad():
            vision = vision.to(device)
            img_inp = processor(images=imgs, return_tensors="pt").to(device)
            dest_img_emb = vision(**img_inp).image_embeds
    img_batch = preprocess_img(imgs, resolution).to(device)
    src_inp = tokenize_prompts(source_prompts, tok, device)
    dst_inp = tokenize_prompts(dest_prompts, tok, device)
    if hp("contrastive_text_loss", False):
        neg_inp = tokenize_prompts(request["negative_prompts"], tok, device)
    src_lookup = [find_token_range(tok, ids, request["source"])[-1] - 1 for ids in src_inp["input_ids"]]
    dst_lookup = [find_token_range(tok, ids, request["dest"])[-1] - 1 for ids in dst_inp["input_ids"]]
    assert len(src_inp["input_ids"]) == len(dst_inp["input_ids"]) == len(img_batch)
    bsz = len(img_batch)
    with torch.no_grad():
        latents = pipe.vae.encode(img_batch).latent_dist.sample() * pipe.vae.config.scaling_factor
        out_d = pipe.text_encoder(**dst_inp)
        dest_repr, dest_pool = out_d[0], out_d[1]
        dest_emb = te_edit.text_projection(dest_pool)
        if objective == "esd" or hp("cal_text_repr_loss", False):
            source_repr = pipe.text_encoder(**src_inp)[0]
            if hp("contrastive_text_loss", False):
                neg_emb = te_edit.text_projection(pipe.text_encoder(**neg_inp)[1])
    fim = None
    if hp("use_ewc", False):
        with np.load(FIM_FILE, allow_pickle=True) as data:
            fim = torch.from_numpy(np.asarray(data["mean.mean"])).to(device)

    def hook(mod, args, out):
        h = _hidden(out)
        if state["source_init"] is None:
            state["source_init"] = h[0, src_lookup[0]].detach().clone()
        for i, idx in enumerate(src_lookup):
            if hp("replace_repr", False):
                h[i, idx, :] = delta
            else:
                h[i, idx, :] += delta
        return out

    handle = get_module(te_edit, hparams["layer_module_tmp"].format(layer)).register_forward_hook(hook)
    try:
        for it in range(hp("v_num_grad_steps")):
            opt.zero_grad()
            noise = torch.randn_like(latents, device=device)
            timesteps = torch.randint(0, noise_scheduler.config.num_train_timesteps, (bsz,), device=device).long()
            noisy = noise_scheduler.add_noise(latents, noise, timesteps)
            out_e = te_edit(**src_inp)
            edit_repr, edit_emb = out_e.last_hidden_state, out_e.text_embeds
            with torch.no_grad():
                if objective == "esd":
                    pred_source = pipe.unet(noisy, timesteps, source_repr).sample
            edit_pred = pipe.unet(noisy, timesteps, edit_repr).sample
            pred_dest = pipe.unet(noisy, timesteps, dest_repr).sample
            source_init = state["source_init"]
            if "ablate" in objective:
                if hp("use_sampled_noise", False):
                    mse = F.mse_loss(noise, edit_pred, reduction="mean")
                else:
                    mse = F.mse_loss(edit_pred, pred_dest, reduction="mean")
                if fim is not None:
                    reg = hp("ewc_lambda", 1e4) * torch.sum(fim * delta ** 2) / (2 * torch.norm(source_init) ** 2)
                else:
                    reg = hp("v_weight_decay") * (torch.norm(delta) / torch.norm(source_init) ** 2)
                loss = mse + reg
            else:
                tmp = pred_dest - hp("esd_mu") * (pred_source - pred_dest)
                loss = F.mse_loss(edit_pred, tmp, reduction="mean") + hp("v_weight_decay") * (torch.norm(delta) / torch.norm(source_init) ** 2)
            if hp("cal_text_repr_loss", False) and not objective == "esd":
                scale = hp("text_repr_loss_scale_factor")
                if hp("contrastive_text_loss", False):
                    single = te_edit.text_projection(pipe.text_encoder(**tokenize_prompts([request["dest"]], tok, device)).pooler_output)
                    emb = torch.cat([single, neg_emb], dim=0)
                    scores = torch.squeeze(-torch.cdist(torch.unsqueeze(edit_emb, dim=0), torch.unsqueeze(emb, dim=0)))
                    loss += scale * (-torch.log_softmax(scores, dim=1)[:, 0].mean(dim=0))
                elif hp("align_object_token", False):
                    loss += scale * F.mse_loss(edit_repr[torch.arange(bsz), src_lookup, :], dest_repr[torch.arange(bsz), dst_lookup, :],
                                               reduction="mean")
                else:
                    loss += scale * F.mse_loss(edit_emb, dest_emb, reduction="mean")
            if request["txt_img_align"]:
                if hp("txt_img_align_loss_metric", "l2") == "cos":
                    align = -(F.cosine_similarity(edit_emb, dest_img_emb, dim=1).mean() - 1)
                elif hp("txt_img_align_loss_metric", "l2") == "l2":
                    align = F.mse_loss(edit_emb, dest_img_emb, reduction="mean")
                else:
                    raise ValueError(f"txt_img_align_loss_metric {hp('txt_img_align_loss_metric')} not supported")
                loss += hp("txt_img_align_scale_factor", 0.0) * align
            loss.backward()
            opt.step()
            max_norm = hp("clamp_norm_factor") * source_init.norm()
            if delta.norm() > max_norm:
                with torch.no_grad():
                    delta[...] = delta * max_norm / delta.norm()
    finally:
        handle.remove()
    return (state["source_init"] + delta).detach()


def compute_z_text_encoder_v2(pipe, request: Dict, hparams: Dict, layer: int, noise_scheduler, resolution: int = 512) -> torch.Tensor:
    """The ``use_new_compute_z`` Stage 1 (compute_z.py:1041-1357), op for op: ``num_edit_tokens`` vectors per concept — the
    last subject token, then (k >= 2) the EOS token and the k - 2 padding positions behind it, the prompts re-tokenized to
    (longest + k - 2) with padding="max_length" (:1074-1083, :1182-1207).  Returns (k, hidden).  What the reference's text
    does and this follows: the weight decay of the ablate objectives is formed under no_grad from float32 copies of the row
    norms (:1277-1281), i.e. it is a constant of the loss; the text term is the MSE over the k looked-up rows for k >= 2, the
    pooled outputs' MSE for k = 1 (:1297-1317); the L2 ball is per row (:1339-1343).  The esd objective and use_ewc read
    ``source_init`` before any assignment (:1275, :1292) — an UnboundLocalError in the reference, NotImplementedError here."""
    from copy import deepcopy
    import torch.nn.functional as F
    hp = lambda k, d=None: hparams.get(k, d)
    objective = hp("objective")
    if objective not in ("ablate-source", "ablate-dest", "esd"):
        raise ValueError(f"Objective {objective} can not be used for compute_z.")
    if objective == "esd" or hp("use_ewc", False):
        raise NotImplementedError("compute_z_text_encoder_v2 reads source_init before assignment for esd / use_ewc (reference :1275, :1292)")
    k = int(hp("num_edit_tokens", 1))
    device = next(pipe.text_encoder.parameters()).device
    te_edit = deepcopy(pipe.text_encoder).to(device)
    tok = pipe.tokenizer
    source_prompts = [p.format(request["source"]) for p in request["prompts"]]
    dest_prompts = [p.format(request["dest"]) for p in request["prompts"]]
    src_inp = tokenize_prompts(source_prompts, tok, device)
    dst_inp = tokenize_prompts(dest_prompts, tok, device)
    n_pad = k - 2
    if k > 1:
        padded = max(len(src_inp["input_ids"][0]), len(dst_inp["input_ids"][0])) + n_pad
        src_inp = tokenize_prompts(source_prompts, tok, device, padding_length=padded)
        dst_inp = tokenize_prompts(dest_prompts, tok, device, padding_length=padded)
    deltas = torch.zeros((k, te_edit.config.hidden_size), requires_grad=True, device=device)
    state = {"inits": None}
    opt = torch.optim.Adam([deltas], lr=hp("v_lr"))
    for m in (te_edit, pipe.vae, pipe.unet, pipe.text_encoder):
        for prm in m.parameters():
            prm.requires_grad = False
    spp = hp("samples_per_prompt", 1)
    if "training_img_paths" in request:
        from PIL import Image
        all_imgs = [Image.open(path) for path in request["training_img_paths"]]
    elif "images" in request:
        all_imgs = request["images"]
    else:
        generator = torch.Generator(device).manual_seed(int(request["seed_train"])) if request["seed_train"] is not None else None
        all_imgs = []
        for _ in range(spp):
            all_imgs.extend(pipe(source_prompts, guidance_scale=7.5, generator=generator).images)
    all_imgs = preprocess_img(all_imgs, resolution)
    bsz = len(source_prompts)
    all_imgs = all_imgs.reshape(spp, bsz, *all_imgs.shape[1:]).transpose(0, 1)       # "(s b) c h w -> b s c h w"
    assert len(all_imgs) % bsz == 0
    src_lookup = [[find_token_range(tok, ids, request["source"])[-1] - 1] for ids in src_inp["input_ids"]]
    dst_lookup = [[find_token_range(tok, ids, request["dest"])[-1] - 1] for ids in dst_inp["input_ids"]]
    if k >= 2:
        src_eos = [int(m.sum()) - 1 for m in src_inp["attention_mask"]]
        dst_eos = [int(m.sum()) - 1 for m in dst_inp["attention_mask"]]
        src_lookup = [lk + list(range(e, e + n_pad + 1)) for lk, e in zip(src_lookup, src_eos)]
        dst_lookup = [lk + list(range(e, e + n_pad + 1)) for lk, e in zip(dst_lookup, dst_eos)]
    assert len(src_inp["input_ids"]) == len(dst_inp["input_ids"]) == len(all_imgs)
    layer_mod = get_module(te_edit, hparams["layer_module_tmp"].format(layer))

    def hook(mod, args, out):
        h = _hidden(out)
        if state["inits"] is None:
            state["inits"] = h[0, src_lookup[0]].detach().clone()          # (k, hidden): rows of the FIRST prompt
        for i, indices in enumerate(src_lookup):
            for j, index in enumerate(indices):
                if hp("replace_repr", False):
                    h[i, index, :] = deltas[j, :]
                else:
                    h[i, index, :] += deltas[j, :]
        return out

    handle = layer_mod.register_forward_hook(hook)
    try:
        for it in range(hp("v_num_grad_steps")):
            opt.zero_grad()
            sample_indices = torch.randint(0, spp, (bsz,))
            img_batch = all_imgs[torch.arange(bsz), sample_indices].to(device)
            with torch.no_grad():
                latents = pipe.vae.encode(img_batch).latent_dist.sample() * pipe.vae.config.scaling_factor
                dest_repr, dest_pool = pipe.text_encoder(**dst_inp)[0:2]
                if hp("cal_text_repr_loss", False):
                    source_repr = pipe.text_encoder(**src_inp)[0]
            noise = torch.randn_like(latents, device=device)
            timesteps = torch.randint(0, noise_scheduler.config.num_train_timesteps, (bsz,), device=device).long()
            noisy = noise_scheduler.add_noise(latents, noise, timesteps)
            edit_repr, edit_pool = te_edit(**src_inp)[0:2]
            if not hp("no_noise_loss", False):
                edit_pred = pipe.unet(noisy, timesteps, edit_repr).sample
                pred_dest = pipe.unet(noisy, timesteps, dest_repr).sample
            inits = state["inits"]
            if hp("use_sampled_noise", False) or request.get("use_real_noise", False):
                mse = F.mse_loss(noise, edit_pred, reduction="mean")
            elif hp("no_noise_loss", False):
                mse = None
            else:
                mse = F.mse_loss(edit_pred, pred_dest, reduction="mean")
            with torch.no_grad():
                delta_norm = torch.Tensor([d_.norm() for d_ in deltas]).mean(0)
                init_norm = torch.Tensor([s_.norm() for s_ in inits]).mean(0)
            reg = hp("v_weight_decay") * (delta_norm / init_norm ** 2)
            loss = reg if hp("no_noise_loss", False) else mse + reg
            if hp("cal_text_repr_loss", False) and request.get("txt_align", True):
                scale = hp("text_repr_loss_scale_factor")
                if k >= 2:
                    edited_rows = torch.stack([edit_repr[i, idx, :] for i, idx in enumerate(src_lookup)], dim=0)
                    wanted_rows = torch.stack([dest_repr[i, idx, :] for i, idx in enumerate(dst_lookup)], dim=0)
                    loss = loss + scale * F.mse_loss(edited_rows, wanted_rows, reduction="mean")
                else:
                    loss = loss + scale * F.mse_loss(edit_pool, dest_pool, reduction="mean")
            loss.backward()
            opt.step()
            for i, s_ in enumerate(inits):
                max_norm = hp("clamp_norm_factor") * s_.norm()
                if deltas[i].norm() > max_norm:
                    with torch.no_grad():
                        deltas[i] = deltas[i] * max_norm / deltas[i].norm()
    finally:
        handle.remove()
    return (deltas + state["inits"]).detach()


# --------------------------------------------------------------------------------------
# second moment + Stage 0 (reference: util/runningstats.py:469-511, 1551-1600;
# dsets/stat_dataset.py:71-172; emcid/layer_stats.py:140-220)
# --------------------------------------------------------------------------------------

class SecondMomentOracle:
    def __init__(self):
        self.count = 0
        self.mom2 = None

    def add(self, a: torch.Tensor):
        if a.dim() != 2:
            a = a.reshape(-1, a.shape[-1])
        if len(a) == 0:
            return
        if self.count == 0:
            self.mom2 = torch.zeros(a.shape[1], a.shape[1], dtype=a.dtype)
        self.count += a.shape[0]
        self.mom2 += a.t().mm(a)

    def moment(self):
        return self.mom2 / self.count


def fixed_random_subset(n_items: int, sample_size: Optional[int], seed: int = 1) -> List[int]:
    """FixedRandomSubsetSampler(dataset, seed=1, end=sample_size) (runningstats.py:1551-1556, 1595-1600)."""
    order = list(range(n_items))
    random.Random(seed).shuffle(order)
    return order[:sample_size]


def length_sorted_subbatches(items: List[List[int]], token_size: int) -> List[List[List[int]]]:
    """length_collation: sort by -len; start a new sub-batch when width*(rows+1) > token_size
    (stat_dataset.py:122-150). Returns lists of token-id rows (unpadded)."""
    items = sorted(items, key=lambda ids: -len(ids))
    groups, cur, width = [], [], 0
    for ids in items:
        if len(ids) == 0:
            break
        if width * (len(cur) + 1) > token_size:
            groups.append(cur)
            cur, width = [], 0
        if not cur:
            width = len(ids)
        cur.append(ids)
    if cur:
        groups.append(cur)
    return groups


def pad_batch(rows: List[List[int]]) -> Dict[str, torch.Tensor]:
    """make_padded_batch: zero-padded ids / position ids / mask (stat_dataset.py:153-163)."""
    w = max(len(r) for r in rows)
    ids = torch.zeros(len(rows), w, dtype=torch.long)
    pos = torch.zeros(len(rows), w, dtype=torch.long)
    msk = torch.zeros(len(rows), w, dtype=torch.long)
    for i, r in enumerate(rows):
        ids[i, :len(r)] = torch.tensor(r)
        pos[i, :len(r)] = torch.arange(len(r))
        msk[i, :len(r)] = 1
    return {"input_ids": ids, "position_ids": pos, "attention_mask": msk}


class _Stop(Exception):
    pass


def layer_stats_text_encoder(model, tokenizer, layer_name: str, captions: List[str], sample_size: Optional[int],
                             batch_tokens: int = 3 * 1024, precision: str = "float32",
                             batch_size: int = 100) -> SecondMomentOracle:
    """Stage 0 for one layer: fixed random caption subset -> groups of 100 -> length-sorted
    sub-batches -> forward stopped at `layer_name` -> attended tokens of its INPUT -> mom2 += a^T a."""
    dtype = getattr(torch, precision)
    stat = SecondMomentOracle()
    order = fixed_random_subset(len(captions), sample_size)
    cap = {}

    def hook(mod, args, out):
        cap["in"] = args[0]
        raise _Stop()

    h = get_module(model, layer_name).register_forward_hook(hook)
    try:
        with torch.no_grad():
            for g in range(0, len(order), batch_size):
                toks = [tokenizer.encode(captions[i], truncation=True, max_length=None) for i in order[g:g + batch_size]]
                for rows in length_sorted_subbatches(toks, batch_tokens):
                    batch = pad_batch(rows)
                    try:
                        model(**batch)
                    except _Stop:
                        pass
                    feats = cap["in"].reshape(-1, cap["in"].shape[-1])[batch["attention_mask"].reshape(-1).nonzero()[:, 0]]
                    stat.add(feats.to(dtype))
    finally:
        h.remove()
    return stat


# --------------------------------------------------------------------------------------
# caches (reference: emcid/emcid_main.py:873-907, 2239-2276; emcid/layer_stats.py:163-174)
# --------------------------------------------------------------------------------------

def stats_path(stat_dir, layer_name, n_samples, precision="float32", model_name="text_encoder",
               ds_name="ccs_filtered", batch_tokens=3 * 1024) -> Path:
    return Path(stat_dir) / f"{model_name}/{ds_name}_stats/{layer_name}_{precision}_mom2_t{batch_tokens}_{n_samples}.npz"


def load_cov(stat_dir, layer_name, n_samples, precision="float32") -> torch.Tensor:
    """C = (mom2 / count).float() from the stats npz (emcid_main.py:2272; runningstats.py:499-510)."""
    with np.load(stats_path(stat_dir, layer_name, n_samples, precision)) as z:
        mom2 = torch.from_numpy(z["mom2.mom2"])
        count = int(z["mom2.count"])
    return (mom2 / count).float()


def load_vstars(cache_name: str, requests, suffix: str = "", use_new_compute_z: bool = False) -> torch.Tensor:
    """zs = stack(v_star, dim=1) -> (hidden, N) fp32 (emcid_main.py:885-899, 977); with ``use_new_compute_z`` the files hold
    (num_edit_tokens, hidden) and zs = "rq num c_i -> c_i (rq num)" of their stack (:972-975)."""
    vs = []
    for r in requests:
        with np.load(Path(cache_name + f"source_{r['source']}_dest_{r['dest']}{suffix}.npz")) as z:
            vs.append(torch.from_numpy(z["v_star"]))
    if use_new_compute_z:
        zs = torch.stack(vs, dim=0)                     # (rq, num, c_i)
        return zs.reshape(-1, zs.shape[-1]).t()
    return torch.stack(vs, dim=1)


# --------------------------------------------------------------------------------------
# Stage 2 (reference: emcid/emcid_main.py:818-1082 SD; :1085-1425 SDXL; :769-815, :38-106 apply)
# --------------------------------------------------------------------------------------

def closed_form_layer(K: torch.Tensor, Zc: torch.Tensor, zs: torch.Tensor, C: torch.Tensor, lam: float,
                      edit_weight: float, layers_left: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """The layer-loop body, emcid_main.py:1016-1050.  K (N,d) fp32, Zc (N,h) fp32, zs (h,N) fp32,
    C (d,d) fp32.  Returns adj_k (d,N) f64, resid (h,N) f64, upd (h,d) f64."""
    ks = K.t()
    sources = zs - Zc.t()
    cov = C * (1 - edit_weight) / 0.5
    s = (edit_weight / 0.5) ** 0.5
    ks64, src64 = ks.double() * s, sources.double() * s
    adj_k = torch.linalg.solve(lam * cov.double() + ks64 @ ks64.T, ks64)
    resid = src64 / layers_left
    upd = resid @ adj_k.T
    return adj_k, resid, upd


def execute_text_encoder(text_encoder, tokenizer, requests, layers: Sequence[int], rewrite_module_tmp: str,
                         zs: torch.Tensor, covs: Dict[int, torch.Tensor], lam: float, edit_weight: float,
                         restore: bool = True, trace: Optional[list] = None, num_edit_tokens: int = 1):
    """Sequential per-layer loop: K fwd, Zc fwd, solve, write W_orig + upd.float() into the live model
    (emcid_main.py:981-1078).  `restore=False` reproduces the SDXL TE2 path (never restored).  ``num_edit_tokens`` > 1: the
    (N, k, .) rows of the num_fact_token branch flattened "rq num c_i -> c_i (rq num)" (:993-1014)."""
    names = [rewrite_module_tmp.format(l) + ".weight" for l in layers]
    weights = {n: get_parameter(text_encoder, n) for n in names}
    backup = {n: w.detach().clone() for n, w in weights.items()}
    deltas = {}
    with torch.no_grad():
        for i, layer in enumerate(layers):
            mod = rewrite_module_tmp.format(layer)
            if num_edit_tokens > 1:
                K, _ = module_input_output_at_words_multi(text_encoder, tokenizer, requests, mod, num_edit_tokens)
                _, Zc = module_input_output_at_words_multi(text_encoder, tokenizer, requests, mod, num_edit_tokens)
                K, Zc = K.reshape(-1, K.shape[-1]), Zc.reshape(-1, Zc.shape[-1])
            else:
                K, _ = module_input_output_at_words(text_encoder, tokenizer, requests, mod)
                _, Zc = module_input_output_at_words(text_encoder, tokenizer, requests, mod)
            adj_k, resid, upd = closed_form_layer(K, Zc, zs, covs[layer], lam, edit_weight, len(layers) - i)
            n = mod + ".weight"
            if upd.shape != weights[n].shape:
                upd = upd.T
            weights[n][...] = backup[n] + upd.float()
            deltas[n] = (adj_k.detach().cpu(), resid.detach().cpu())
            if trace is not None:
                trace.append({"K": K.clone(), "Zc": Zc.clone(), "adj_k": adj_k, "resid": resid})
        if restore:
            for n, w in weights.items():
                w[...] = backup[n]
    return deltas


def insert_deltas(text_encoder, deltas):
    """w += (adj_k @ resid^T) matched to w.shape, .float() (emcid_main.py:802-809, 2279-2298)."""
    with torch.no_grad():
        for n, (adj_k, resid) in deltas.items():
            w = get_parameter(text_encoder, n)
            upd = adj_k @ resid.T
            if upd.shape != w.shape:
                upd = upd.T
            w[...] += upd.float()


def apply_emcid_to_text_encoder(pipe, requests, hparams: Dict, mom2_weight=None, edit_weight=None,
                                cache_name=None, stats_dir=None, trace=None):
    """emcid_main.py:769-815 with all v* and C pre-cached on disk.  `hparams` is a plain dict
    (the JSON fields); it is mutated in place like the reference does (:846-847)."""
    hparams["mom2_update_weight"] = mom2_weight if mom2_weight is not None else hparams["mom2_update_weight"]
    hparams["edit_weight"] = edit_weight if edit_weight is not None else hparams.get("edit_weight", 0.5)
    requests = copy.deepcopy(requests)
    tmpl = hparams["rewrite_module_tmp"]
    zs = load_vstars(cache_name, requests, use_new_compute_z=hparams.get("use_new_compute_z", False))
    covs = {l: load_cov(stats_dir, tmpl.format(l), hparams["mom2_n_samples"], hparams["mom2_dtype"])
            for l in hparams["layers"]}
    deltas = execute_text_encoder(pipe.text_encoder, pipe.tokenizer, requests, hparams["layers"], tmpl, zs, covs,
                                  hparams["mom2_update_weight"], hparams["edit_weight"], trace=trace,
                                  num_edit_tokens=int(hparams.get("num_edit_tokens", 1)))
    insert_deltas(pipe.text_encoder, deltas)
    return pipe, deltas


def apply_emcid_to_sdxl_text_encoders(pipe, requests, hparams: Dict, mom2_weight=None, mom2_weight_2=None,
                                      edit_weight=None, cache_name=None, stat_dir=None, stat_dir_2=None):
    """emcid_main.py:38-106 + :1085-1425.  TE1 is restored before the deltas are inserted; TE2 is NOT
    (reference :1410 vs :93-99), so TE2 ends at W + upd.float() + upd'.float() — reproduced as is."""
    hparams["mom2_update_weight"] = mom2_weight if mom2_weight is not None else hparams["mom2_update_weight"]
    hparams["mom2_update_weight_2"] = mom2_weight_2 if mom2_weight_2 is not None else hparams["mom2_update_weight_2"]
    hparams["edit_weight"] = edit_weight if edit_weight is not None else hparams.get("edit_weight", 0.5)
    requests = copy.deepcopy(requests)
    tmpl = hparams["rewrite_module_tmp"]
    zs = load_vstars(cache_name, requests)
    zs2 = load_vstars(cache_name, requests, "_2")
    covs = {l: load_cov(stat_dir, tmpl.format(l), hparams["mom2_n_samples"], hparams["mom2_dtype"]) for l in hparams["layers"]}
    covs2 = {l: load_cov(stat_dir_2, tmpl.format(l), hparams["mom2_n_samples"], hparams["mom2_dtype"]) for l in hparams["layers_2"]}
    d1 = execute_text_encoder(pipe.text_encoder, pipe.tokenizer, requests, hparams["layers"], tmpl, zs, covs,
                              hparams["mom2_update_weight"], hparams["edit_weight"], restore=True)
    d2 = execute_text_encoder(pipe.text_encoder_2, pipe.tokenizer_2, requests, hparams["layers_2"], tmpl, zs2, covs2,
                              hparams["mom2_update_weight_2"], hparams["edit_weight"], restore=False)
    insert_deltas(pipe.text_encoder, d1)
    insert_deltas(pipe.text_encoder_2, d2)
    return pipe, d1, d2


# --------------------------------------------------------------------------------------
# Cross-attention K/V edit (reference: emcid/emcid_main.py:314-548; emcid/compute_ks.py:52-141;
# emcid/layer_stats.py:333-427, :470-495, :555-575; util/globals.py:37-38)
# --------------------------------------------------------------------------------------

UNET_EDIT_TEMPLATES = {   # util/globals.py:37-38 (the two templates this path uses)
    "cross-k": "{}.{}.attentions.{}.transformer_blocks.0.attn2.to_k",
    "cross-v": "{}.{}.attentions.{}.transformer_blocks.0.attn2.to_v",
}


def get_to_edit_layername_unet(template_key, block_type, block_idx, sub_idx) -> str:
    """layer_stats.py:555-575 restricted to the cross-k / cross-v templates."""
    name = UNET_EDIT_TEMPLATES[template_key].format(block_type, block_idx, sub_idx)
    if "mid_block" in block_type:
        name = name.replace(f"mid_block.{block_idx}.", "mid_block.")
    return name


def get_all_cross_attn_kv_layer_names(unet) -> List[str]:
    """layer_stats.py:470-495: block types in this order, per block every cross-k then every cross-v, names that do
    not resolve on the UNet are skipped."""
    names = []
    for block_type, count in (("down_blocks", 4), ("up_blocks", 4), ("mid_block", 1)):
        for idx in range(count):
            for key in ("cross-k", "cross-v"):
                for sub in (0, 1, 2):
                    name = get_to_edit_layername_unet(key, block_type, idx, sub)
                    obj = unet
                    try:
                        for part in name.split("."):
                            obj = getattr(obj, part)
                    except AttributeError:
                        continue
                    names.append(name)
    return names


def layers_input_output_at_words_cross_attn(pipe, requests, module_names) -> Tuple[Dict, Dict]:
    """compute_ks.py:52-141.  Text encoder on all N*P prompts; per request one UNet forward on its P prompts'
    embeddings with hooks on every to_k / to_v; input / output rows at each prompt's last subject token, mean over
    the request's prompts.  Returns ({name: (N, hidden)}, {name: (N, out)})."""
    device = pipe.device
    prompts, subjects, counts = expand_requests(requests)
    assert len(set(counts)) == 1, "All the requests should have the same number of prompts."
    batch_size = counts[0]
    inp = tokenize_prompts(prompts, pipe.tokenizer, device)
    lookup = [find_token_range(pipe.tokenizer, ids, w)[-1] - 1 for ids, w in zip(inp["input_ids"], subjects)]
    with torch.no_grad():
        rep = pipe.text_encoder(**inp)[0]                       # last hidden state (final LayerNorm applied)
    latents = torch.zeros(batch_size, pipe.unet.config.in_channels, pipe.unet.config.sample_size,
                          pipe.unet.config.sample_size, device=device)     # dummies: the projections only see `rep`
    timesteps = torch.zeros(batch_size, dtype=torch.long, device=device)
    mods = dict(pipe.unet.named_modules())
    ins = {n: [] for n in module_names}
    outs = {n: [] for n in module_names}
    for b in range(rep.shape[0] // batch_size):
        cap = {}
        hooks = [mods[n].register_forward_hook(lambda m, a, o, n=n: cap.__setitem__(n, (a[0], o))) for n in module_names]
        try:
            with torch.no_grad():
                pipe.unet(latents, timesteps, encoder_hidden_states=rep[b * batch_size:(b + 1) * batch_size])
        finally:
            for h in hooks:
                h.remove()
        idx = lookup[b * batch_size:(b + 1) * batch_size]
        for n in module_names:
            ins[n].append(torch.stack([cap[n][0][i, j, :] for i, j in enumerate(idx)], 0).detach().clone().mean(0))
            outs[n].append(torch.stack([cap[n][1][i, j, :] for i, j in enumerate(idx)], 0).detach().clone().mean(0))
    return {n: torch.stack(v, 0) for n, v in ins.items()}, {n: torch.stack(v, 0) for n, v in outs.items()}


def layer_stats_cross_attn_kv(pipe, layer_name: str, captions: List[str], sample_size: Optional[int],
                              batch_tokens: int = 3 * 1024, precision: str = "float32",
                              batch_size: int = 4) -> SecondMomentOracle:
    """layer_stats.py:333-427: fixed random caption subset -> groups of FOUR (:350) -> length-sorted sub-batches ->
    text encoder -> UNet on dummy latents stopped at `layer_name` -> attended tokens of its INPUT -> mom2 += a^T a."""
    dtype = getattr(torch, precision)
    stat = SecondMomentOracle()
    order = fixed_random_subset(len(captions), sample_size)
    cap = {}

    def hook(mod, args, out):
        cap["in"] = args[0]
        raise _Stop()

    h = dict(pipe.unet.named_modules())[layer_name].register_forward_hook(hook)
    latents = torch.zeros(batch_size, pipe.unet.config.in_channels, pipe.unet.config.sample_size, pipe.unet.config.sample_size)
    timesteps = torch.zeros(batch_size, dtype=torch.long)
    try:
        with torch.no_grad():
            for g in range(0, len(order), batch_size):
                toks = [pipe.tokenizer.encode(captions[i], truncation=True, max_length=None) for i in order[g:g + batch_size]]
                for rows in length_sorted_subbatches(toks, batch_tokens):
                    batch = pad_batch(rows)
                    rep = pipe.text_encoder(**batch).last_hidden_state
                    try:
                        pipe.unet(latents, timesteps, encoder_hidden_states=rep)
                    except _Stop:
                        pass
                    feats = cap["in"].reshape(-1, cap["in"].shape[-1])[batch["attention_mask"].reshape(-1).nonzero()[:, 0]]
                    stat.add(feats.to(dtype))
    finally:
        h.remove()
    return stat


def compute_z_unet_x_kv(pipe, request: Dict, hparams: Dict, noise_scheduler, resolution: int = 512) -> Dict[str, torch.Tensor]:
    """Stage 1 of the cross-attention sibling (compute_z.py:2407-2645), op for op: a deep copy of the UNet whose attn2.to_k /
    to_v outputs get their delta added in place at every prompt's last subject token (the TraceDict edit_output of :2482-2492,
    here forward hooks on the copy), per step one VAE encode + sample, the clean UNet's source / unconditional (/ safety)
    predictions, the safe-latent-diffusion or esd supervision, MSE + mean weight decay, one Adam over all deltas, projection of
    every delta onto its L2 ball.  Randomness: the global torch generators, consumed in the reference's order."""
    from copy import deepcopy
    import torch.nn.functional as F
    hp = lambda k, d=None: hparams.get(k, d)
    device = pipe.device
    unet_to_edit = deepcopy(pipe.unet).to(device)
    tok = pipe.tokenizer
    source_prompts = [p.format(request["source"]) for p in request["prompts"]]
    src_inp = tokenize_prompts(source_prompts, tok, device)
    src_lookup = [find_token_range(tok, ids, request["source"])[-1] - 1 for ids in src_inp["input_ids"]]
    if hp("sld_supervision", False):
        if hp("all_safe", False):
            safe_words = ('hate, harassment, violence, suffering, humiliation, harm, suicide, '
                          'sexual, nudity, bodily fluids, blood, obscene gestures, illegal activity, '
                          'drug use, theft, vandalism, weapons, child abuse, brutality, cruelty')
        else:
            safe_words = request["safe words"]
        if hp("sld_type", "max") == "max":
            sld = {"sld_guidance_scale": 5000, "sld_warmup_steps": 0, "sld_threshold": 1.0, "sld_momentum_scale": 0.5, "sld_mom_beta": 0.7}
        elif hp("sld_type") == "strong":
            sld = {"sld_guidance_scale": 2000, "sld_warmup_steps": 7, "sld_threshold": 0.025, "sld_momentum_scale": 0.5, "sld_mom_beta": 0.7}
        else:
            raise ValueError(f"sld_type {hp('sld_type')} not supported")
        sld = {k: torch.tensor(v).to(device) for k, v in sld.items()}
    layer_names = get_all_cross_attn_kv_layer_names(pipe.unet)
    delta_dict = {n: torch.zeros((get_module(pipe.unet, n).out_features,), requires_grad=True, device=device) for n in layer_names}
    init_dict = {n: None for n in layer_names}

    def make_hook(name):
        def hook(mod, args, cur_out):
            if init_dict[name] is None:
                init_dict[name] = cur_out[0, src_lookup[0]].detach().clone()
            for i, idx in enumerate(src_lookup):
                if hp("replace_repr", False):
                    cur_out[i, idx, :] = delta_dict[name]
                else:
                    cur_out[i, idx, :] += delta_dict[name]
            return cur_out
        return hook

    opt = torch.optim.Adam([delta_dict[n] for n in layer_names], lr=hp("v_lr"))
    for m in (unet_to_edit, pipe.vae, pipe.unet, pipe.text_encoder):
        for prm in m.parameters():
            prm.requires_grad = False
    spp = hp("samples_per_prompt", 1)
    if "images" in request:                    # (not in the reference: lets a GPU run use the images a CPU run generated)
        all_imgs = request["images"]
    else:
        generator = torch.Generator(pipe.device).manual_seed(int(request["seed_train"]))
        all_imgs = []
        for _ in range(spp):
            all_imgs.extend(pipe(source_prompts, guidance_scale=7.5, generator=generator).images)
    bsz = len(source_prompts)
    assert len(all_imgs) % bsz == 0
    all_imgs = preprocess_img(all_imgs, resolution)
    all_imgs = all_imgs.reshape(spp, bsz, *all_imgs.shape[1:]).transpose(0, 1)       # "(s b) c h w -> b s c h w"
    with torch.no_grad():
        source_repr = pipe.text_encoder(**src_inp)[0]
        if hp("sld_supervision", False):
            safe_repr = pipe.text_encoder(**tokenize_prompts([safe_words] * bsz, tok, device))[0]
        uncond_repr = pipe.text_encoder(**tokenize_prompts([""] * bsz, tok, device))[0]
    copy_mods = dict(unet_to_edit.named_modules())
    for it in range(hp("v_num_grad_steps")):
        opt.zero_grad()
        sample_indices = torch.randint(0, spp, (bsz,))
        imgs = all_imgs[torch.arange(bsz), sample_indices].to(device)
        with torch.no_grad():
            latents = pipe.vae.encode(imgs).latent_dist.sample()
            latents = latents * pipe.vae.config.scaling_factor
        handles = [copy_mods[n].register_forward_hook(make_hook(n)) for n in layer_names]
        try:
            noise = torch.randn_like(latents, device=device)
            timesteps = torch.randint(0, noise_scheduler.config.num_train_timesteps, (bsz,), device=device).long()
            noisy = noise_scheduler.add_noise(latents, noise, timesteps)
            with torch.no_grad():
                pred_source = pipe.unet(noisy, timesteps, source_repr).sample
                pred_uncond = pipe.unet(noisy, timesteps, uncond_repr).sample
                if hp("sld_supervision", False):
                    pred_safety = pipe.unet(noisy, timesteps, safe_repr).sample
                    scale = torch.clamp(torch.abs((pred_source - pred_safety)) * sld["sld_guidance_scale"], max=1.0)
                    concept_scale = torch.where((pred_source - pred_safety) >= sld["sld_threshold"], torch.zeros_like(scale), scale)
                    guidance = torch.mul((pred_safety - pred_uncond), concept_scale)
                    supervision = pred_source - guidance
                elif hp("objective") == "esd":
                    supervision = pred_uncond - hp("esd_mu") * (pred_source - pred_uncond)
            edit_pred = unet_to_edit(noisy, timesteps, source_repr).sample
            mse = F.mse_loss(edit_pred, supervision, reduction="mean")
            weight_decay = 0
            for n in layer_names:
                weight_decay += hp("v_weight_decay") * (torch.norm(delta_dict[n]) / torch.norm(init_dict[n]) ** 2)
            loss = mse + weight_decay / len(layer_names)
            loss.backward()
            opt.step()
            for n in layer_names:
                max_norm = hp("clamp_norm_factor") * init_dict[n].norm()
                if delta_dict[n].norm() > max_norm:
                    with torch.no_grad():
                        delta_dict[n][...] = delta_dict[n] * max_norm / delta_dict[n].norm()
        finally:
            for hd in handles:
                hd.remove()
    with torch.no_grad():
        return {n: init_dict[n] + delta_dict[n] for n in layer_names}


def load_cov_cross_attn(stats_dir, layer_name, n_samples, precision="float32") -> torch.Tensor:
    """emcid_main.py:2217-2232 with the statistics already cached on disk (layer_stats.py:361: model_name "unet")."""
    with np.load(stats_path(stats_dir, layer_name, n_samples, precision, model_name="unet")) as z:
        return (torch.from_numpy(z["mom2.mom2"]) / int(z["mom2.count"])).float()


def load_vstars_cross_attn(cache_name: str, requests, layer_names) -> Dict[str, torch.Tensor]:
    """emcid_main.py:373-391 + :424-426: per request one npz whose entries are pickled {"v_star": array};
    zs[name] = stack(dim=1) -> (out, N)."""
    per = {n: [] for n in layer_names}
    for r in requests:
        data = np.load(Path(cache_name + f"source_{r['source']}.npz"), allow_pickle=True)
        for n in layer_names:
            per[n].append(torch.from_numpy(data[n].item()["v_star"]))
    return {n: torch.stack(v, dim=1) for n, v in per.items()}


def execute_cross_attn(pipe, requests, hparams: Dict, cache_name, stats_dir, mom2_weight=None, edit_weight=None,
                       trace: Optional[dict] = None):
    """emcid_main.py:314-508.  Every to_k / to_v gets the closed form with the SAME keys (the text embedding at the
    subject token), its own targets and statistics, resid NOT divided by a layer count (:473).  The UNet is restored."""
    hparams["mom2_update_weight"] = mom2_weight if mom2_weight is not None else hparams["mom2_update_weight"]
    hparams["edit_weight"] = edit_weight if edit_weight is not None else hparams["edit_weight"]
    requests = copy.deepcopy(requests)
    names = get_all_cross_attn_kv_layer_names(pipe.unet)
    params = dict(pipe.unet.named_parameters())
    weights = {f"{n}.weight": params[f"{n}.weight"] for n in names}
    backup = {k: v.detach().clone() for k, v in weights.items()}
    zs = load_vstars_cross_attn(cache_name, requests, names)
    deltas = {}
    with torch.no_grad():
        ks, cur = layers_input_output_at_words_cross_attn(pipe, requests, names)
        for n in names:
            C = load_cov_cross_attn(stats_dir, n, hparams["mom2_n_samples"], hparams["mom2_dtype"])
            adj_k, resid, upd = closed_form_layer(ks[n], cur[n], zs[n].to(ks[n].device), C, hparams["mom2_update_weight"],
                                                  hparams["edit_weight"], 1)
            w = weights[f"{n}.weight"]
            if upd.shape != w.shape:
                upd = upd.T
            w[...] = backup[f"{n}.weight"] + upd.float()
            deltas[f"{n}.weight"] = (adj_k.detach().cpu(), resid.detach().cpu())
            if trace is not None:
                trace[n] = {"K": ks[n].clone(), "Zc": cur[n].clone()}
        for k, v in weights.items():
            v[...] = backup[k]
    return deltas


def apply_emcid_to_cross_attn(pipe, requests, hparams: Dict, cache_name, stats_dir, mom2_weight=None, edit_weight=None,
                              trace=None):
    """emcid_main.py:511-548: w += float(adj_k @ resid^T matched to w.shape)."""
    deltas = execute_cross_attn(pipe, requests, hparams, cache_name, stats_dir, mom2_weight, edit_weight, trace)
    insert_deltas(pipe.unet, deltas)
    return pipe, deltas


# --------------------------------------------------------------------------------------
# UCE closed form, the baseline the reference ships beside EMCID (emcid/uce_train.py:31-213 text-encoder fc2
# variant, :216-416 cross-attention K/V variant).  W_new = (lam W + sum v k^T)(lam I + sum k k^T)^-1.
# `dtype` = the arithmetic after the (always fp32) encoder forward: float32 restates the reference op for op
# (incl. torch.inverse); float64 is the same algebra with the rounding of the fp32 inverse removed, which is what the
# HIP path (fp64 Cholesky) is compared against.
# --------------------------------------------------------------------------------------

def _uce_texts(old_text_, new_text_, retain_text_):
    """uce_train.py:52-66 / :268-282: '' as a new text becomes ' '; no retain list -> ['']."""
    old_texts = list(old_text_)
    new_texts = [(' ' if t == '' else t) for t in new_text_]
    ret_texts = [''] if retain_text_ is None else list(retain_text_)
    return old_texts, new_texts, ret_texts


def _uce_tokens(tokenizer, texts):
    return tokenizer(list(texts), padding="max_length", max_length=tokenizer.model_max_length, truncation=True,
                     return_tensors="pt")


def _uce_row_windows(attention_mask, n_rows: int):
    """uce_train.py:109-127 / :304-316: rows from the last subject token to the end, the longer text's tail cut
    so both have the same number of rows."""
    f_old = int(attention_mask[0].sum().item()) - 2
    f_new = int(attention_mask[1].sum().item()) - 2
    far = max(f_old, f_new)
    return (f_old, n_rows - max(0, far - f_old)), (f_new, n_rows - max(0, far - f_new))


def _uce_value(layer_fn, old_rows, new_rows, technique):
    """uce_train.py:155-169: 'tensor' removes from the new value its component along the (whole-matrix normalised)
    old value; anything else takes the new value."""
    if technique == 'tensor':
        u = layer_fn(old_rows)
        u = u / u.norm()
        new_embs = layer_fn(new_rows)
        return new_embs - (u * new_embs).sum() * u
    return layer_fn(new_rows)


def _uce_outer_sums(value, context):
    """uce_train.py:170-174: sum_r v_r k_r^T and sum_r k_r k_r^T — as the reference forms them, one outer product per
    row and a sum over rows (a GEMM would round differently, and the fp32 inverse amplifies that)."""
    cv = context.reshape(context.shape[0], context.shape[1], 1)
    cvt = context.reshape(context.shape[0], 1, context.shape[1])
    vv = value.reshape(value.shape[0], value.shape[1], 1)
    return (vv @ cvt).sum(dim=0), (cv @ cvt).sum(dim=0)


def edit_text_encoder_uce(pipe, old_text_, new_text_, retain_text_, layer_to_edit=11, lamb=0.1, erase_scale=0.1,
                          preserve_scale=0.1, technique='tensor', dtype=torch.float32) -> torch.Tensor:
    """uce_train.py:31-213.  Quirks kept: the retain pass sits INSIDE the loop over edits (:178), so it is added once per
    edit; values come from the whole fc2 module, bias included (:158, :163); only the weight is replaced."""
    module = get_module(pipe.text_encoder, f"text_model.encoder.layers.{layer_to_edit}.mlp.fc2")
    old_texts, new_texts, ret_texts = _uce_texts(old_text_, new_text_, retain_text_)
    W = module.weight.detach().to(dtype)
    b = module.bias.detach().to(dtype)
    layer_fn = lambda x: torch.nn.functional.linear(x, W, b)

    def fc2_inputs(texts):
        cap = {}
        h = module.register_forward_hook(lambda m, a, o: cap.__setitem__("x", a[0].detach().clone()))
        try:
            ti = _uce_tokens(pipe.tokenizer, texts)
            with torch.no_grad():
                pipe.text_encoder(ti.input_ids.to(pipe.device))
        finally:
            h.remove()
        return ti, cap["x"].to(dtype)

    mat1 = lamb * W.clone()
    mat2 = lamb * torch.eye(W.shape[1], dtype=dtype, device=W.device)
    for old_text, new_text in zip(old_texts, new_texts):
        ti, x = fc2_inputs([old_text, new_text])
        (o0, o1), (n0, n1) = _uce_row_windows(ti.attention_mask, x.shape[1])
        context, new_rows = x[0, o0:o1], x[1, n0:n1]
        m1, m2 = _uce_outer_sums(_uce_value(layer_fn, context, new_rows, technique), context)
        mat1 += erase_scale * m1
        mat2 += erase_scale * m2
        for t in ret_texts:
            _, x = fc2_inputs([t, t])
            m1, m2 = _uce_outer_sums(layer_fn(x[1]), x[0])
            mat1 += preserve_scale * m1
            mat2 += preserve_scale * m2
    new_w = mat1 @ torch.inverse(mat2)
    with torch.no_grad():
        module.weight.copy_(new_w.to(module.weight.dtype))
    return new_w


def edit_model_uce(pipe, old_text_, new_text_, retain_text_, layers_to_edit=None, lamb=0.1, erase_scale=0.1,
                   preserve_scale=0.1, with_to_k=True, technique='tensor', dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """uce_train.py:216-416.  context = final text embeddings (:298); the retain pass is outside the loop over edits
    here (:392).  The reference's projection list (:232-247) is built from the K/V layer-name list with the
    `.to_k`/`.to_v` suffix stripped, so every attention block appears in it TWICE: 2x16 to_v entries, then (with_to_k)
    2x16 to_k entries, and `layers_to_edit` indexes that doubled list (:286).  Its "reset" loop (:254-260) re-attaches a
    fresh copy per entry, which leaves the FIRST entry of each pair pointing at a detached module: only an edit through
    the second entry reaches the UNet, and it starts from the original weight."""
    blocks = [n.replace('.to_k', '').replace('.to_v', '') for n in get_all_cross_attn_kv_layer_names(pipe.unet)]
    entries = [n + '.to_v' for n in blocks] + ([n + '.to_k' for n in blocks] if with_to_k else [])
    attached = [pn not in entries[j + 1:len(blocks) * (1 + j // len(blocks))] for j, pn in enumerate(entries)]
    proj_names = [pn if att else None for pn, att in zip(entries, attached)]
    mods = dict(pipe.unet.named_modules())
    old_texts, new_texts, ret_texts = _uce_texts(old_text_, new_text_, retain_text_)

    def embed(texts):
        ti = _uce_tokens(pipe.tokenizer, texts)
        with torch.no_grad():
            return ti, pipe.text_encoder(ti.input_ids.to(pipe.device))[0].to(dtype)

    out = {}
    for layer_num, pname in enumerate(proj_names):
        if (layers_to_edit is not None and layer_num not in layers_to_edit) or pname is None:
            continue
        module = mods[pname]
        W = module.weight.detach().to(dtype)
        layer_fn = lambda x: torch.nn.functional.linear(x, W)
        mat1 = lamb * W.clone()
        mat2 = lamb * torch.eye(W.shape[1], dtype=dtype, device=W.device)
        for old_text, new_text in zip(old_texts, new_texts):
            ti, emb = embed([old_text, new_text])
            (o0, o1), (n0, n1) = _uce_row_windows(ti.attention_mask, emb.shape[1])
            context, new_rows = emb[0, o0:o1], emb[1, n0:n1]
            m1, m2 = _uce_outer_sums(_uce_value(layer_fn, context, new_rows, technique), context)
            mat1 += erase_scale * m1
            mat2 += erase_scale * m2
        for t in ret_texts:
            _, emb = embed([t, t])
            m1, m2 = _uce_outer_sums(layer_fn(emb[1]), emb[0])
            mat1 += preserve_scale * m1
            mat2 += preserve_scale * m2
        out[pname] = mat1 @ torch.inverse(mat2)
    with torch.no_grad():
        for pname, new_w in out.items():
            mods[pname].weight.copy_(new_w.to(mods[pname].weight.dtype))
    return out
