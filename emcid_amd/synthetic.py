"""Deterministic synthetic inputs for tests, golden-vector generation and bench.py.

No network, no CLIP vocabulary and no Stable Diffusion weights are available
offline, so every benchmark/test input is synthetic:

* tokenizer   - an in-memory ``CLIPTokenizer`` whose BPE vocabulary is
                characters + whole-word merges for the prompt templates, so a
                prompt such as ``"painting by c0042"`` is ~8 tokens (real CLIP
                BPE gives a similar count for artist names).
* encoder     - a seeded random-init ``CLIPTextModel`` at toy or real dims
                (SD-v1.4 / SDXL-TE1: 768/3072/12 layers; SDXL-TE2: 1280/5120/32).
* requests    - the reference's request dict schema
                (reference: emcid/emcid_main.py:832-840, test_examples/*.json).
* v* cache    - one ``v_star`` npz per request in the reference's path scheme
                (reference: emcid/emcid_main.py:873-890, 951-968).
* stats cache - ``mom2`` npz per layer in the reference's scheme
                (reference: emcid/layer_stats.py:166-174).
"""
from __future__ import annotations

import json
import os
import string
from pathlib import Path
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

TEMPLATE_WORDS = [
    "painting", "by", "artwork", "style", "of", "the", "in", "paint", "a", "an",
    "photo", "image", "picture", "with", "and", "on", "at", "is", "famous",
    "portrait", "landscape", "drawing", "sketch", "realist", "artist",
]
ARTIST_TEMPLATES = ["painting by {}", "artwork by {}", "style of {}"]

ENCODER_DIMS = {
    # name: (hidden, intermediate, layers, heads, act)
    "toy": (32, 128, 5, 2, "quick_gelu"),
    "toy2": (48, 192, 7, 3, "gelu"),
    "sd-v1.4": (768, 3072, 12, 12, "quick_gelu"),
    "sdxl-te1": (768, 3072, 12, 12, "quick_gelu"),
    "sdxl-te2": (1280, 5120, 32, 20, "gelu"),
}


CONSONANTS = "bcdfghjklmnprstvwz"
VOWELS = "aeiou"
SYLLABLES = [c + v for c in CONSONANTS for v in VOWELS]          # 90 consonant-vowel syllables


def synthetic_vocab(words: Sequence[str] = TEMPLATE_WORDS, syllables=False
                    ) -> Tuple[Dict[str, int], List[Tuple[str, str]]]:
    """Character vocabulary plus left-to-right merge chains for ``words``.  ``syllables=True`` (the benchmark
    vocabulary) also merges every consonant-vowel pair, so a name such as ``kazumi`` is 3 tokens — the token
    count real CLIP BPE gives a typical artist name (the golden fixtures keep the plain vocabulary).
    ``syllables="wide"`` adds the 1 620 consonant-vowel-consonant syllables behind those (``kaz``, ``kaz</w>``): enough
    distinct one-token words for name lists with the first-word statistics of real artist lists (``artist_names``)."""
    chars = list(string.ascii_lowercase + string.digits + ".,'-")
    vocab: Dict[str, int] = {}
    for c in chars:
        vocab[c] = len(vocab)
    for c in chars:
        vocab[c + "</w>"] = len(vocab)
    merges: List[Tuple[str, str]] = []
    for w in words:
        syms = list(w[:-1]) + [w[-1] + "</w>"]
        cur = syms[0]
        for s in syms[1:]:
            if (cur, s) not in merges:
                merges.append((cur, s))
            cur = cur + s
            if cur not in vocab:
                vocab[cur] = len(vocab)
    if syllables:   # lower priority than the word chains above, so template words still become one token
        for c in CONSONANTS:
            for v in VOWELS:
                for tail in ("", "</w>"):
                    if (c, v + tail) not in merges:
                        merges.append((c, v + tail))
                    if c + v + tail not in vocab:
                        vocab[c + v + tail] = len(vocab)
    if syllables == "wide":      # consonant-vowel-consonant, ranked behind every consonant-vowel merge
        for c in CONSONANTS:
            for v in VOWELS:
                for c2 in CONSONANTS:
                    for tail in ("", "</w>"):
                        if (c + v, c2 + tail) not in merges:
                            merges.append((c + v, c2 + tail))
                        if c + v + c2 + tail not in vocab:
                            vocab[c + v + c2 + tail] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    return vocab, merges


def artist_names(n: int, seed: int = 3) -> List[str]:
    """n distinct two-word names with the shape of the reference's artist lists (data/artists/info/erased-1000artists-….txt:
    1 000 names, 86 % of two words, 655 distinct first words, the commonest one 21 times, ~14 characters): a first word that is
    ONE token of the ``syllables="wide"`` vocabulary (a consonant-vowel-consonant syllable, drawn from a Zipf-like pool: common
    first names are single CLIP BPE tokens and are shared) and a last word of 2-4 tokens nobody shares — 3-5 tokens per name.
    Synthetic: the reference's list itself is not shipped."""
    rng = np.random.default_rng(seed)
    cvc = [c + v + c2 for c in CONSONANTS for v in VOWELS for c2 in CONSONANTS]
    pool = [cvc[i] for i in rng.permutation(len(cvc))]
    # a quarter of the names take one of 60 common first words (p ~ 1 / (rank + 3)), the rest any of the others: on n = 1000
    # ~650 distinct first words, the commonest ~21, 15, 12, 12, 11, 11 times (the reference's list: 655; 21, 12, 11, 11, 9, 9)
    head = 1.0 / (np.arange(1, 61, dtype=np.float64) + 3.0)
    p = np.concatenate([0.25 * head / head.sum(), np.full(len(pool) - 60, 0.75 / (len(pool) - 60))])
    names, seen = [], set()
    while len(names) < n:
        first = pool[int(rng.choice(len(pool), p=p))]
        k = int(rng.choice([2, 3, 4], p=[0.45, 0.4, 0.15]))
        parts = []
        for _ in range(k):
            parts.append(SYLLABLES[int(rng.integers(0, len(SYLLABLES)))] if rng.random() < 0.7 else cvc[int(rng.integers(0, len(cvc)))])
        nm = first + " " + "".join(parts)
        if nm not in seen:
            seen.add(nm)
            names.append(nm)
    return names


def syllable_names(n: int, seed: int = 3, syllables_per_name: int = 3) -> List[str]:
    """n distinct pronounceable names (``kazumi``), deterministic."""
    rng = np.random.default_rng(seed)
    names, seen = [], set()
    while len(names) < n:
        nm = "".join(SYLLABLES[i] for i in rng.integers(0, len(SYLLABLES), size=syllables_per_name))
        if nm not in seen:
            seen.add(nm)
            names.append(nm)
    return names


def build_tokenizer(vocab: Optional[Dict[str, int]] = None, merges=None, model_max_length: int = 77):
    from transformers import CLIPTokenizer

    if vocab is None:
        vocab, merges = synthetic_vocab()
    merges = [tuple(m) for m in merges]
    return CLIPTokenizer(vocab=dict(vocab), merges=merges, model_max_length=model_max_length)


def add_trained_like_outliers(model, seed: int = 5) -> None:
    """In place: give a random-init CLIP text encoder the TENSOR statistics a trained one is known for and a Gaussian init lacks,
    the way trained networks carry them — large activations met by small weights, so that the function stays well-conditioned
    (a gain of 10^3 in front of Gaussian q / k projections would saturate every softmax: the fp32 reference itself is then
    only good to 5e-3, measured) while the intermediate tensors are heavy-tailed:
      * six hidden channels whose LayerNorm gains are 10^2-10^3 times the median (biases off zero), with the matching input
        columns of q | k | v (LN1) and fc1 (LN2) scaled down by the same factor (up to a jitter of 1-2): every LayerNorm
        output row has a few entries a thousand times the rest — the case a per-row scale is worst at;
      * two "massive activation" channels of the residual stream (a constant 25-40 in the position embeddings);
      * eight heavy fc1 rows (x 10-40, biases shifted) with their fc2 columns scaled down: heavy-tailed rows of the fc2 input;
      * two heads per layer that lean on the start token (a bias on their queries, keys x 2).
    Seeded, so the reference and the product build the same model."""
    g = torch.Generator().manual_seed(seed)
    root = getattr(model, "text_model", model)
    h = root.final_layer_norm.weight.numel()
    n_out = 6
    chans = torch.randperm(h, generator=g)[:n_out]
    with torch.no_grad():
        root.embeddings.position_embedding.weight[:, chans[:2]] += 25.0 + 15.0 * torch.rand(2, generator=g)
        for layer in root.encoder.layers:
            mlp, at = layer.mlp, layer.self_attn
            for ln, consumers in ((layer.layer_norm1, (at.q_proj, at.k_proj, at.v_proj)), (layer.layer_norm2, (mlp.fc1,))):
                m = 10.0 ** (2.0 + torch.rand(n_out, generator=g))            # 10^2 .. 10^3
                ln.weight[chans] *= m
                ln.bias[chans] += (torch.rand(n_out, generator=g) - 0.5) * 2.0 * m.sqrt()
                for lin in consumers:
                    lin.weight[:, chans] *= (1.0 + torch.rand(n_out, generator=g)) / m
            rows = torch.randperm(mlp.fc1.out_features, generator=g)[:8]
            r = 10.0 + 30.0 * torch.rand(8, generator=g)
            mlp.fc1.weight[rows] *= r[:, None]
            mlp.fc1.bias[rows] += torch.randn(8, generator=g)
            mlp.fc2.weight[:, rows] /= r[None, :]
            hd = h // at.num_heads
            for head in torch.randperm(at.num_heads, generator=g)[:2].tolist():
                at.k_proj.weight[head * hd:(head + 1) * hd] *= 2.0
                at.q_proj.bias[head * hd:(head + 1) * hd] += 0.5


def build_text_encoder(kind: str = "toy", vocab_size: Optional[int] = None, seed: int = 0,
                       name_or_path: str = "synthetic/clip-text", projection_dim: Optional[int] = None, outliers: bool = False):
    """Seeded random-init CLIPTextModel (fp32, eval, no grad); with ``projection_dim`` a CLIPTextModelWithProjection (SDXL's
    second encoder: ``.text_embeds``, what Stage 1 of the SDXL pair reads); ``outliers``: ``add_trained_like_outliers``."""
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection

    hidden, inter, layers, heads, act = ENCODER_DIMS[kind]
    if vocab_size is None:
        vocab_size = len(synthetic_vocab()[0])
    cfg = CLIPTextConfig(
        hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers,
        num_attention_heads=heads, vocab_size=vocab_size, max_position_embeddings=77,
        hidden_act=act, bos_token_id=vocab_size - 2, eos_token_id=vocab_size - 1,
        pad_token_id=vocab_size - 1, **({"projection_dim": projection_dim} if projection_dim else {}),
    )
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    model = (CLIPTextModelWithProjection(cfg) if projection_dim else CLIPTextModel(cfg)).eval()
    torch.random.set_rng_state(gen_state)
    for p in model.parameters():
        p.requires_grad_(False)
    if outliers:
        add_trained_like_outliers(model, seed=seed + 5)
    model.config._name_or_path = name_or_path
    return model


class SyntheticPipe(SimpleNamespace):
    """Duck-typed stand-in for a diffusers pipeline: exactly the attributes the
    edit path touches (.text_encoder, .tokenizer, .device,
    [.text_encoder_2, .tokenizer_2])."""

    @property
    def device(self):
        return next(self.text_encoder.parameters()).device

    def __call__(self, prompts, guidance_scale: float = 7.5, generator=None, **kw):
        """Stand-in for image generation (the reference samples Stage-1 training images with ``pipe(prompts, ...).images``,
        emcid/compute_z.py:2503-2510): one deterministic image per prompt, seeded by a draw from ``generator``."""
        res = int(getattr(self, "image_resolution", 32))
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,), generator=generator,
                                 device=generator.device if generator is not None else "cpu").item())
        return SimpleNamespace(images=make_images(len(prompts), res, seed=seed))

    def to(self, device):
        self.text_encoder.to(device)
        if getattr(self, "text_encoder_2", None) is not None:
            self.text_encoder_2.to(device)
        if getattr(self, "unet", None) is not None:
            self.unet.to(device)
        if getattr(self, "vae", None) is not None:
            self.vae.to(device)
        return self


def build_pipe(kind: str = "toy", device: str = "cpu", sdxl: bool = False, seed: int = 0,
               syllables=False, projection_dim: Optional[int] = None, outliers: bool = False) -> SyntheticPipe:
    vocab, merges = synthetic_vocab(syllables=syllables)
    tok = build_tokenizer(vocab, merges)
    if not sdxl:
        te = build_text_encoder(kind, len(vocab), seed=seed, outliers=outliers)
        return SyntheticPipe(text_encoder=te.to(device), tokenizer=tok)
    kind1, kind2 = ("toy", "toy2") if kind.startswith("toy") else ("sdxl-te1", "sdxl-te2")
    te1 = build_text_encoder(kind1, len(vocab), seed=seed, name_or_path="synthetic/clip-text-1")
    te2 = build_text_encoder(kind2, len(vocab), seed=seed + 1, name_or_path="synthetic/clip-text-2", projection_dim=projection_dim)
    return SyntheticPipe(text_encoder=te1.to(device), tokenizer=tok,
                         text_encoder_2=te2.to(device), tokenizer_2=build_tokenizer(vocab, merges))


def build_clip_towers(pipe: "SyntheticPipe", projection_dim: int = 16, seed: int = 7, image_size: int = 32):
    """Stand-ins for what the reference's compute_z_text_encoder_v1 fetches from the hub (openai/clip-vit-large-patch14,
    emcid/compute_z.py:1376-1378, :1440): (CLIPTextModelWithProjection whose text tower carries ``pipe.text_encoder``'s weights —
    the hub checkpoint's text tower IS SD-v1.x's encoder —, a small seeded CLIPVisionModelWithProjection, a CLIPProcessor for
    ``image_size`` images).  Offline: built from configs, no files."""
    from transformers import (CLIPImageProcessor, CLIPProcessor, CLIPTextConfig, CLIPTextModelWithProjection, CLIPVisionConfig,
                              CLIPVisionModelWithProjection)
    te = pipe.text_encoder
    cfg = te.config.to_dict()
    cfg["projection_dim"] = projection_dim
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    text = CLIPTextModelWithProjection(CLIPTextConfig(**cfg)).eval()
    hidden = int(te.config.hidden_size)
    vision = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=hidden, intermediate_size=2 * hidden, num_hidden_layers=2,
                                                            num_attention_heads=2, image_size=image_size, patch_size=8,
                                                            projection_dim=projection_dim)).eval()
    torch.random.set_rng_state(gen_state)
    inner = text.text_model if hasattr(text, "text_model") else text
    src = te.text_model if hasattr(te, "text_model") else te
    inner.load_state_dict({k: v.detach().cpu() for k, v in src.state_dict().items()}, strict=True)
    for m in (text, vision):
        for prm in m.parameters():
            prm.requires_grad_(False)
    proc = CLIPProcessor(image_processor=CLIPImageProcessor(size={"shortest_edge": image_size},
                                                            crop_size={"height": image_size, "width": image_size}),
                         tokenizer=pipe.tokenizer)
    return text, vision, proc


# ---- cross-attention K/V stand-in for the UNet -------------------------------------------------------------------
# Channel widths of the 16 cross-attention blocks: SD-v1.4 [external: its UNet config], and a toy set.  The tree of
# module NAMES is what the reference addresses (util/globals.py:37-38 through emcid/layer_stats.py:470-495).
UNET_BLOCK_CHANNELS = {"sd-v1.4": (320, 640, 1280, 1280), "toy": (16, 24, 48, 48)}   # none equal to a text hidden size: see emcid_main._edit_cross_attn


class _CrossAttention(torch.nn.Module):
    def __init__(self, text_hidden, channels):
        super().__init__()
        self.to_k = torch.nn.Linear(text_hidden, channels, bias=False)
        self.to_v = torch.nn.Linear(text_hidden, channels, bias=False)


class _TransformerBlock(torch.nn.Module):
    def __init__(self, text_hidden, channels):
        super().__init__()
        self.attn2 = _CrossAttention(text_hidden, channels)


class _Transformer2D(torch.nn.Module):
    def __init__(self, text_hidden, channels):
        super().__init__()
        self.transformer_blocks = torch.nn.ModuleList([_TransformerBlock(text_hidden, channels)])


class _UNetBlock(torch.nn.Module):
    def __init__(self, text_hidden, channels, n_attn):
        super().__init__()
        if n_attn:
            self.attentions = torch.nn.ModuleList([_Transformer2D(text_hidden, channels) for _ in range(n_attn)])


class SyntheticUNet(torch.nn.Module):
    """The part of a diffusers UNet2DConditionModel the cross-attention edit touches: the ``attn2.to_k`` / ``to_v``
    projections of the text embedding, under the real module names (down 0-2: two each, down 3: none, mid: one,
    up 0: none, up 1-3: three each — 16 blocks, 32 matrices for SD-v1.x).  ``forward`` feeds
    ``encoder_hidden_states`` through every projection, so hooks on them see what they would see in a real UNet."""

    def __init__(self, text_hidden: int, kind: str = "toy", seed: int = 11):
        super().__init__()
        c = UNET_BLOCK_CHANNELS[kind]
        st = torch.random.get_rng_state()
        torch.manual_seed(seed)
        self.down_blocks = torch.nn.ModuleList([_UNetBlock(text_hidden, c[i], 2 if i < 3 else 0) for i in range(4)])
        self.mid_block = _UNetBlock(text_hidden, c[3], 1)
        self.up_blocks = torch.nn.ModuleList([_UNetBlock(text_hidden, c[3 - i], 3 if i > 0 else 0) for i in range(4)])
        torch.random.set_rng_state(st)
        for p in self.parameters():
            p.requires_grad_(False)
        self.config = SimpleNamespace(_name_or_path=f"synthetic/unet-{kind}", in_channels=4, sample_size=8)

    def forward(self, sample, timestep, encoder_hidden_states=None, **kw):
        # every projection sees the text embedding (what the reference's hooks record); the returned "noise prediction"
        # depends smoothly on the projections' outputs and on the timestep, so Stage 1 has something to differentiate
        s = 0.0
        for m in self.modules():
            if isinstance(m, _CrossAttention):
                k = m.to_k(encoder_hidden_states)
                v = m.to_v(encoder_hidden_states)
                s = s + (k.mean(dim=(1, 2)) * v.mean(dim=(1, 2)))
        if torch.is_tensor(s):
            t = torch.as_tensor(timestep, dtype=sample.dtype, device=sample.device).reshape(-1)
            g = (1.0 + 0.5 * torch.tanh(8.0 * s) + 0.1 * torch.sin(t / 160.0)).reshape(-1, *([1] * (sample.dim() - 1)))
            sample = sample * g + 3.0 * s.reshape(-1, *([1] * (sample.dim() - 1)))
        return SimpleNamespace(sample=sample)


class _LatentDist:
    def __init__(self, mean, logvar):
        self.mean, self.std = mean, torch.exp(0.5 * logvar)

    def sample(self, generator=None):
        # like diffusers' DiagonalGaussianDistribution.sample (randn_tensor): one draw from the global generator of the
        # mean's device, or from `generator` on ITS device (a CPU generator serves a device tensor) moved over afterwards
        dev = self.mean.device if generator is None else generator.device
        eps = torch.randn(self.mean.shape, generator=generator, device=dev, dtype=self.mean.dtype)
        return self.mean + self.std * eps.to(self.mean.device)


class SyntheticVAE(torch.nn.Module):
    """The part of a diffusers AutoencoderKL Stage 1 touches: ``encode(x).latent_dist.sample()`` and
    ``config.scaling_factor``; an 8x-downsampling conv to 2 x 4 latent channels (mean | log-variance)."""

    def __init__(self, seed: int = 21):
        super().__init__()
        st = torch.random.get_rng_state()
        torch.manual_seed(seed)
        self.conv = torch.nn.Conv2d(3, 8, kernel_size=8, stride=8)
        torch.random.set_rng_state(st)
        for p in self.parameters():
            p.requires_grad_(False)
        self.config = SimpleNamespace(scaling_factor=0.18215)

    def encode(self, x):
        mom = self.conv(x)
        return SimpleNamespace(latent_dist=_LatentDist(mom[:, :4], mom[:, 4:].clamp(-30.0, 20.0) - 4.0))


class DDPMNoiseSchedule:
    """``add_noise`` of the DDPM training schedule Stable Diffusion v1.x ships in its ``scheduler`` folder [external:
    CompVis/stable-diffusion-v1-4 scheduler_config.json: 1000 steps, scaled_linear betas 0.00085 .. 0.012] — all Stage 1
    uses of ``DDPMScheduler`` (reference: emcid/compute_z.py:378, :521-524):  x_t = sqrt(acp_t) x_0 + sqrt(1 - acp_t) eps."""

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps)
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)

    @classmethod
    def from_pretrained(cls, *args, **kwargs):
        return cls()

    def add_noise(self, original_samples, noise, timesteps):
        acp = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        t = timesteps.to(original_samples.device)
        a = acp[t].sqrt().reshape(-1, *([1] * (original_samples.dim() - 1)))
        b = (1.0 - acp[t]).sqrt().reshape(-1, *([1] * (original_samples.dim() - 1)))
        return a * original_samples + b * noise


def make_images(n: int, resolution: int, seed: int = 31):
    """n deterministic RGB PIL images (smooth random fields) of ``resolution`` x ``resolution``."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    imgs = []
    for _ in range(n):
        low = rng.random((4, 4, 3))
        arr = np.kron(low, np.ones((resolution // 4, resolution // 4, 1)))
        imgs.append(Image.fromarray((arr * 255).astype(np.uint8), "RGB"))
    return imgs


def add_diffusion(pipe: "SyntheticPipe", kind: str = "toy") -> "SyntheticPipe":
    """UNet stand-in (cross-attention projections under their real names, differentiable output), VAE stand-in and the
    scheduler attribute: everything Stage 1 (compute_z_text_encoder) reads from a pipeline besides the text encoder."""
    add_unet(pipe, kind)
    pipe.vae = SyntheticVAE().to(pipe.device)
    return pipe


class SyntheticUNetXL(SyntheticUNet):
    """SyntheticUNet with SDXL's extra conditioning: ``forward(sample, timestep, encoder_hidden_states=, added_cond_kwargs=
    {"text_embeds", "time_ids"})``; the output also depends (smoothly) on the pooled text embedding and the size ids."""

    def __init__(self, text_hidden: int, add_dim: int, kind: str = "toy", seed: int = 13):
        super().__init__(text_hidden, kind, seed)
        st = torch.random.get_rng_state()
        torch.manual_seed(seed + 1)
        self.add_embedding = torch.nn.Linear(add_dim + 6, 8)
        torch.random.set_rng_state(st)
        for p in self.parameters():
            p.requires_grad_(False)

    def forward(self, sample, timestep, encoder_hidden_states=None, added_cond_kwargs=None, **kw):
        out = super().forward(sample, timestep, encoder_hidden_states).sample
        if added_cond_kwargs is not None:
            cond = torch.cat([added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"].to(sample.dtype) / 1024.0], dim=-1)
            a = torch.tanh(self.add_embedding(cond)).mean(dim=1)
            out = out * (1.0 + 0.2 * a).reshape(-1, *([1] * (sample.dim() - 1)))
        return SimpleNamespace(sample=out)


def add_sdxl_diffusion(pipe: "SyntheticPipe", kind: str = "toy", sample_size: int = 4) -> "SyntheticPipe":
    """What Stage 1 of the SDXL pair (compute_z_sdxl_text_encoders) reads from a StableDiffusionXLPipeline besides the two text
    encoders: the UNet with ``added_cond_kwargs``, the VAE, ``scheduler`` (add_noise), ``default_sample_size``,
    ``vae_scale_factor`` and ``_get_add_time_ids``.  ``pipe.text_encoder_2`` must carry a projection (``build_pipe(...,
    projection_dim=...)``)."""
    h1, h2 = pipe.text_encoder.config.hidden_size, pipe.text_encoder_2.config.hidden_size
    dev = pipe.device
    pipe.unet = SyntheticUNetXL(h1 + h2, pipe.text_encoder_2.config.projection_dim, kind).to(dev)
    pipe.vae = SyntheticVAE().to(dev)
    pipe.scheduler = DDPMNoiseSchedule()
    pipe.default_sample_size, pipe.vae_scale_factor = sample_size, 8

    def _get_add_time_ids(original_size, crops_coords_top_left, target_size=None, dtype=None, text_encoder_projection_dim=None,
                          source_size=None):
        # diffusers: list(original_size + crops_coords_top_left + target_size) as one row (the reference's pinned version names
        # the last argument source_size, emcid/compute_z.py:868-874)
        last = target_size if target_size is not None else source_size
        return torch.tensor([list(original_size) + list(crops_coords_top_left) + list(last)], dtype=dtype)

    pipe._get_add_time_ids = _get_add_time_ids
    return pipe


def add_unet(pipe: "SyntheticPipe", kind: str = "toy", seed: int = 11) -> "SyntheticPipe":
    """Attach a SyntheticUNet (+ the scheduler attribute the reference reads for its dummy timesteps) to a pipe."""
    hidden = pipe.text_encoder.config.hidden_size
    dev = pipe.device
    pipe.unet = SyntheticUNet(hidden, kind, seed).to(dev)
    pipe.scheduler = SimpleNamespace(config=SimpleNamespace(num_train_timesteps=1000))
    return pipe


def xattn_vstar_cache_path(cache_name: str, request: Dict) -> Path:
    # reference: emcid/emcid_main.py:373-377
    return Path(cache_name + f"source_{request['source']}.npz")


def write_xattn_vstar_cache(cache_name: str, requests: Sequence[Dict], layer_dims: Dict[str, int], seed: int = 1,
                            scale: float = 1.0) -> Dict[str, np.ndarray]:
    """One npz per request in the reference's format (emcid_main.py:411-420): key = layer name, value = a pickled
    ``{"v_star": array}``.  Returns {layer_name: (N, out_dim)}."""
    rng = np.random.default_rng(seed)
    vs = {ln: (rng.standard_normal((len(requests), dim)) * scale).astype(np.float32) for ln, dim in layer_dims.items()}
    for i, r in enumerate(requests):
        p = xattn_vstar_cache_path(cache_name, r)
        p.parent.mkdir(parents=True, exist_ok=True)
        np.savez(p, **{ln: {"v_star": vs[ln][i]} for ln in layer_dims})
    return vs


def make_requests(n: int, dest: str = "a realist artist", templates: Sequence[str] = ARTIST_TEMPLATES,
                  seed_train: int = 2024, ragged: bool = False, names: str = "index", name_seed: int = 3) -> List[Dict]:
    """n unique synthetic concepts in the reference's request schema: ``c0000`` … (``names="index"``) or
    3-syllable names (``names="syllable"``, needs the ``syllables=True`` vocabulary to tokenize compactly; ``name_seed``
    draws another set of names: 729 000 possible, so two seeds share about one name in a thousand).
    ``ragged`` gives requests differing prompt counts (1..len(templates))."""
    reqs = []
    sources = (syllable_names(n, seed=name_seed) if names == "syllable" else
               artist_names(n, seed=name_seed) if names == "artist" else [f"c{i:04d}" for i in range(n)])
    for i in range(n):
        k = len(templates) if not ragged else 1 + (i % len(templates))
        reqs.append({
            "source": sources[i],
            "dest": dest,
            "prompts": list(templates[:k]),
            "seed_train": seed_train,
        })
    return reqs


def own_prompt_requests(reqs: Sequence[Dict], seed: int = 977) -> List[Dict]:
    """The same requests with prompts that share NOTHING but the start token: every request brings its own three prompts, each
    with three 3-syllable words of its own (>= 9 tokens on the ``syllables=True`` vocabulary) in front of ``by {}`` — the
    prefix trie then has (almost) one row per token.  Deterministic in (len(reqs), seed)."""
    n = len(reqs)
    words = syllable_names(9 * n, seed=seed)
    return [dict(r, prompts=[" ".join(words[9 * i + 3 * p:9 * i + 3 * p + 3]) + " by {}" for p in range(3)])
            for i, r in enumerate(reqs)]


def vstar_cache_path(cache_name: str, request: Dict, suffix: str = "") -> Path:
    # reference: emcid/emcid_main.py:885-890 (SD) and :1157-1166 (SDXL "_2")
    return Path(cache_name + f"source_{request['source']}_dest_{request['dest']}{suffix}.npz")


def write_vstar_cache(cache_name: str, requests: Sequence[Dict], hidden: int, seed: int = 1,
                      suffix: str = "", scale: float = 1.0) -> np.ndarray:
    """Writes one ``v_star`` npz per request and returns the (N, hidden) array."""
    rng = np.random.default_rng(seed)
    vs = (rng.standard_normal((len(requests), hidden)) * scale).astype(np.float32)
    for v, r in zip(vs, requests):
        p = vstar_cache_path(cache_name, r, suffix)
        p.parent.mkdir(parents=True, exist_ok=True)
        np.savez(p, v_star=v)
    return vs


def stats_file(stats_dir, layer_name: str, n_samples: int, model_name: str = "text_encoder",
               ds_name: str = "ccs_filtered", precision: str = "float32", batch_tokens: int = 3 * 1024) -> Path:
    # reference: emcid/layer_stats.py:163-174
    return Path(stats_dir) / f"{model_name}/{ds_name}_stats/{layer_name}_{precision}_mom2_t{batch_tokens}_{n_samples}.npz"


def synthetic_second_moment(d: int, seed: int, t: int = 4096, anisotropy: float = 3.0) -> Tuple[np.ndarray, int]:
    """mom2 = X^T X for X ~ N(0, diag(s)^2) with log-uniform column scales, fp32."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(t, d, generator=g, dtype=torch.float32)
    scales = torch.exp(torch.linspace(0.0, -anisotropy, d))[torch.randperm(d, generator=g)]
    x = x * scales
    return (x.t() @ x).numpy(), t


def write_stats_cache(stats_dir, layer_names: Sequence[str], d: int, n_samples: int, seed: int = 2,
                      t: int = 4096, **kw) -> Dict[str, np.ndarray]:
    """Writes a reference-format stats npz per layer (keys probed in SURVEY.md §5:
    mom2.constructor, mom2.count, mom2.mom2, sample_size). Returns {layer_name: C}."""
    out = {}
    for i, ln in enumerate(layer_names):
        mom2, count = synthetic_second_moment(d, seed + i, t=t)
        f = stats_file(stats_dir, ln, n_samples, **kw)
        f.parent.mkdir(parents=True, exist_ok=True)
        np.savez(f, **{
            "mom2.constructor": "util.runningstats.SecondMoment()",
            "mom2.count": count,
            "mom2.mom2": mom2,
            "sample_size": n_samples,
        })
        out[ln] = mom2 / count
    return out


def make_captions(n: int, seed: int = 2, mean_words: float = 9.0) -> List[Dict[str, str]]:
    """Synthetic caption set in the reference's JSON shape [{"caption": ...}]
    (reference: dsets/stat_dataset.py:90-92)."""
    rng = np.random.default_rng(seed)
    words = TEMPLATE_WORDS
    caps = []
    for _ in range(n):
        k = int(np.clip(rng.lognormal(np.log(mean_words), 0.45), 1, 60))
        ws = [words[j] for j in rng.integers(0, len(words), size=k)]
        if rng.random() < 0.5:
            ws.append("c%04d" % rng.integers(0, 10000))
        caps.append({"caption": " ".join(ws)})
    return caps


def write_captions(path, n: int, seed: int = 2) -> List[Dict[str, str]]:
    caps = make_captions(n, seed)
    os.makedirs(os.path.dirname(str(path)) or ".", exist_ok=True)
    with open(path, "w") as f:
        json.dump(caps, f)
    return caps


def sd_hparams_dict(layers=(7, 8, 9, 10), mom2_update_weight: int = 4000, edit_weight: float = 0.5,
                    mom2_n_samples: int = 100000, prefix: str = "text_model.") -> Dict:
    """Field-for-field the shipped ``ly-7-11`` JSON (reference:
    hparams/dest_s-200_c-1.5_ly-7-11_lr-0.2_wd-5e-04_txt-align-0.01.json) with overridable layers."""
    return {
        "layers": list(layers), "clamp_norm_factor": 1.5, "layer_selection": "all",
        "fact_token": "subject_last", "v_num_grad_steps": 100, "v_lr": 0.2, "v_weight_decay": 5e-4,
        "mom2_adjustment": True, "mom2_update_weight": mom2_update_weight,
        "rewrite_module_tmp": prefix + "encoder.layers.{}.mlp.fc2",
        "layer_module_tmp": prefix + "encoder.layers.{}",
        "mlp_module_tmp": prefix + "encoder.layers.{}.mlp",
        "attn_module_tmp": prefix + "encoder.layers.{}.self_attn",
        "ln_f_module": prefix + "final_layer_norm",
        "mom2_dataset": "ccs_filtered", "mom2_n_samples": mom2_n_samples, "mom2_dtype": "float32",
        "objective": "ablate-dest", "esd_mu": "None", "cal_text_repr_loss": True,
        "text_repr_loss_scale_factor": 0.01, "edit_weight": edit_weight,
    }


def sdxl_hparams_dict(layers=(8, 9, 10), layers_2=(26, 27, 28, 29, 30), mom2_update_weight: int = 4000,
                      mom2_update_weight_2: int = 10000, edit_weight: float = 0.5,
                      mom2_n_samples: int = 100000, prefix: str = "text_model.") -> Dict:
    """Field-for-field the shipped SDXL JSON (reference:
    hparams/sdxl-dest_s-100_c-1.2_ly-8-11_ly2-26-31_lr-0.1_wd-8e-03_txt-align-0.01.json)."""
    d = sd_hparams_dict(layers, mom2_update_weight, edit_weight, mom2_n_samples, prefix)
    d.update({"layers_2": list(layers_2), "mom2_update_weight_2": mom2_update_weight_2,
              "clamp_norm_factor": 1.2, "v_lr": 0.1, "v_weight_decay": 8e-3,
              "text_repr_loss_scale_factor": 0.005})
    return d
