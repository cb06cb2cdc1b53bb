"""K/Z assembly: fc2 input ("key") and fc2 output at each concept's last subject token.

Host-side counterpart of the slice of the reference's emcid/compute_z.py that Stage 2 calls:
``tokenize_prompts`` (:56-74) and ``get_module_input_output_at_words`` (:2252-2384) — and Stage 1 itself,
``compute_z_text_encoder`` (:315-649, the per-concept Adam optimisation of v* through the UNet),
which emcid_main calls on a v* cache miss when the pipeline carries a UNet and a VAE.

MI355X-first differences, results identical:
* the prompt batch (ids, mask, lookup index per prompt, request segment offsets) is built once on the host
  and lives in HBM (``PromptBatch``); the reference re-tokenizes and re-searches on every call;
* the N*P Python indexings and N Python means are one gather+mean kernel (csrc/gram_f32.hip,
  ``emcid_gather_mean_f32``), summed in prompt order with a true division so it is bit-compatible with
  torch-CPU's ``.mean(0)``;
* the forward stops at the hooked module (the reference runs the remaining layers and discards them).
"""
import operator
import os
import weakref
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import hip, host_text
from .causal_trace import TokenRangeFinder
from .clip_attention import hip_attention
from .nethook import StopForward, get_module


def tokenize_prompts(prompts, tokenizer, device, padding_length=None):
    if padding_length is None:
        enc = tokenizer(prompts, return_tensors="pt", padding=True, truncation=True)
    else:
        enc = tokenizer(prompts, return_tensors="pt", padding="max_length", truncation=True, max_length=padding_length)
    return {k: v.to(device) for k, v in enc.items()}


def expand_request_prompts(requests: Sequence[Dict], first: Optional[Dict] = None) -> Tuple[List[str], List[str], List[int]]:
    """Flattened prompt strings, the subject of each, and prompts-per-request.
    ``source_prompts`` (pre-formatted) wins over ``prompts`` when the FIRST request carries it, and the
    per-request count is taken from ``prompts`` when the first request has that key — both as the reference
    does (compute_z.py:2270-2283, :2318-2320).  ``first``: the first request of the WHOLE list when ``requests`` is a slice."""
    first = requests[0] if first is None else first
    use_pre = "source_prompts" in first
    count_key = "prompts" if "prompts" in first else "source_prompts"
    prompts, subjects, counts = [], [], []
    for r in requests:
        ps = list(r["source_prompts"]) if use_pre else [p.format(r["source"]) for p in r["prompts"]]
        prompts.extend(ps)
        subjects.extend([r["source"]] * len(ps))
        counts.append(len(r[count_key]))
    return prompts, subjects, counts


@dataclass
class PromptBatch:
    """Device-resident tokenized prompts of one encoder for a request list."""
    inputs: Dict[str, torch.Tensor]      # input_ids, attention_mask (B, S) in HBM
    lookup: torch.Tensor                 # (B,) int64: position of the last subject token of each prompt
    seg: torch.Tensor                    # (N+1,) int64: prompt offsets per request
    n_requests: int
    lookup_host: List[int]
    ids_host: Optional[np.ndarray] = None   # (B, S) int64 copy of input_ids on the host (the trie is built from it)
    lookup_multi: Optional[torch.Tensor] = None   # (k, B) int64, num_edit_tokens = k > 1: [last subject token, EOS, EOS + 1, ...] per prompt

    @property
    def n_prompts(self):
        return int(self.lookup.numel())


def tokenize_lists(tokenizer, prompts: Sequence[str]) -> Dict[str, np.ndarray]:
    """``tokenizer(prompts, padding=True, truncation=True)`` as (B, S) int64 arrays (reference: compute_z.py:65).

    Three routes, all returning what the public call returns:
    1. ``host_text.NativeClipBpe`` — the C++ restatement of the CLIP pipeline in ``libemcid_host.so``, built from this
       tokenizer's own vocabulary and merges and checked against it on a probe set when first seen; prompts it does not
       serve (non-ASCII, special-token syntax) go through the HF tokenizer row by row.  The longest prompt is also encoded
       with the public call and compared, every time.
    2. For a tokenizers-backed HF tokenizer without a native twin, a ONE-prompt public call configures the backend's
       truncation and padding exactly as transformers does for these arguments (``set_truncation_and_padding``), then the
       backend encodes the batch without offsets (``encode_batch_fast``), checked against the one-prompt result.
    3. The public call."""
    bt = getattr(tokenizer, "_tokenizer", None)
    if bt is not None and len(prompts) > 8:
        longest = max(range(len(prompts)), key=lambda i: len(prompts[i]))
        twin = host_text.NativeClipBpe.for_tokenizer(tokenizer)
        if twin is not None:
            out = twin.tokenize(tokenizer, prompts)
            probe = tokenizer([prompts[longest]], padding=True, truncation=True)
            want = probe["input_ids"][0]
            ids, mask = out["input_ids"], out["attention_mask"]
            if ids.shape[1] >= len(want) and ids[longest, :len(want)].tolist() == want \
                    and int(mask[longest].sum()) == int(sum(probe["attention_mask"][0])):
                return out
            host_text.NativeClipBpe.disable(tokenizer)      # never trust a twin that disagreed once
        if hasattr(bt, "encode_batch_fast"):
            try:
                probe = tokenizer([prompts[longest]], padding=True, truncation=True)
                encs = bt.encode_batch_fast(list(prompts), add_special_tokens=True)
                ids = np.array([e.ids for e in encs], dtype=np.int64)
                mask = np.array([e.attention_mask for e in encs], dtype=np.int64)
                want = probe["input_ids"][0]
                if ids.ndim == 2 and ids.shape[1] >= len(want) and ids[longest, :len(want)].tolist() == want \
                        and int(mask[longest].sum()) == int(sum(probe["attention_mask"][0])) \
                        and set(probe.keys()) == {"input_ids", "attention_mask"}:
                    return {"input_ids": ids, "attention_mask": mask}
            except Exception:
                pass
    enc = tokenizer(list(prompts), padding=True, truncation=True)
    return {k: np.asarray(v, dtype=np.int64) for k, v in enc.items()}


_FINDERS = weakref.WeakKeyDictionary()     # tokenizer -> TokenRangeFinder (its per-token decode table is a pure function of the tokenizer)


def finder_for(tokenizer) -> TokenRangeFinder:
    try:
        f = _FINDERS.get(tokenizer)
        if f is None:
            f = _FINDERS[tokenizer] = TokenRangeFinder(tokenizer)
        return f
    except TypeError:       # not weak-referenceable
        return TokenRangeFinder(tokenizer)


def build_prompt_batch(tokenizer, requests: Sequence[Dict], device, finder: Optional[TokenRangeFinder] = None,
                       truncate: bool = True) -> PromptBatch:
    prompts, subjects, counts = expand_request_prompts(requests)
    enc = tokenize_lists(tokenizer, prompts)
    finder = finder or finder_for(tokenizer)
    ids_host = enc["input_ids"]
    lookup = [r[-1] - 1 for r in finder.batch(ids_host, subjects)]
    if len(ids_host) != len(lookup):
        raise ValueError("The number of prompts and lookup indices should be the same.")
    S = ids_host.shape[1]
    lk = np.asarray(lookup, dtype=np.int64)
    bad = np.nonzero((lk < 0) | (lk >= S))[0]
    if bad.size:
        i = int(bad[0])
        raise ValueError(f"lookup index {lookup[i]} outside the padded prompt (S={S}) for prompt {prompts[i]!r}")
    seg = np.cumsum([0] + counts)
    if seg[-1] != len(prompts):
        raise ValueError(f"request prompt counts ({seg[-1]}) do not cover the {len(prompts)} prompts")
    if truncate:
        # CLIP text attention is causal: nothing at or before a lookup token depends on later positions, so
        # the columns after the last lookup index (EOS, padding) are never needed by the K/Z gather.
        keep = int(lk.max()) + 1
        enc = {k: np.ascontiguousarray(v[:, :keep]) for k, v in enc.items()}
    return PromptBatch(
        inputs={k: torch.from_numpy(v).to(device) for k, v in enc.items()},
        lookup=torch.from_numpy(lk).to(device),
        seg=torch.from_numpy(seg.astype(np.int64)).to(device),
        n_requests=len(requests), lookup_host=lookup, ids_host=enc["input_ids"])


@dataclass
class PromptChunk:
    """Host-side tokenization of a contiguous slice of the request list (prompts truncated behind the last lookup token)."""
    ids: np.ndarray          # (B_c, S_c) int64
    lookup: Sequence[int]    # position of the last subject token per prompt (list or int64 array)
    counts: List[int]        # prompts per request
    n_requests: int
    verify: Optional[Callable[[], bool]] = None      # deferred tokenizer cross-check (templated_prompt_chunk), None = already done
    offsets: Optional[np.ndarray] = None             # (n_requests + 1,) int64 prompt offsets of the requests, when already known

    def request_offsets(self) -> np.ndarray:
        if self.offsets is None:
            self.offsets = np.concatenate([[0], np.cumsum(np.asarray(self.counts, dtype=np.int64))]).astype(np.int64)
        return self.offsets


_GET_SOURCE, _GET_PROMPTS = operator.itemgetter("source"), operator.itemgetter("prompts")


def templated_prompt_chunk(tokenizer, requests: Sequence[Dict], first: Dict, defer_probe: bool = False) -> Optional[PromptChunk]:
    """The (few templates) x (many names) shape of a mass edit WITHOUT building, joining and re-splitting the prompt strings:
    ``p.format(source)`` (reference compute_z.py:2278-2283) for templates with exactly one ``{}`` and no other brace is
    ``prefix + source + suffix``, so ``libemcid_host`` encodes every distinct prefix, suffix and source once
    (``emcid_bpe_encode_templated``) and walks each row for its subject with the subjects passed once
    (``emcid_find_token_ranges_idx``).  Same ids and lookup positions as the generic path (tests/test_host_cpu.py); returns
    None whenever the request list is outside that shape (pre-formatted ``source_prompts``, other format fields, a
    tokenizer without a native twin, a non-string source) and the caller takes the generic path.  ``defer_probe``: the
    per-call comparison of the longest row against the public tokenizer call (a never-seen prompt: ~0.1 ms of the HF
    tokenizer) is handed back as ``chunk.verify`` instead of being run here — the engine calls it after it has launched the
    leading layers and redoes the preparation on the generic path if it ever says no."""
    if "source_prompts" in first or "prompts" not in first or len(requests) < 3 or getattr(tokenizer, "_tokenizer", None) is None:
        return None
    twin = host_text.NativeClipBpe.for_tokenizer(tokenizer)
    if twin is None:
        return None
    try:
        names = list(map(_GET_SOURCE, requests))                 # (C-level iteration: this runs before the GPU has anything to do)
        keys = list(map(tuple, map(_GET_PROMPTS, requests)))
        distinct = dict.fromkeys(keys)                   # the distinct template tuples, in order of first appearance
    except TypeError:                                    # unhashable prompt entries, requests that are not mappings
        return None
    if set(map(type, names)) != {str}:                   # format() would str() anything else
        return None
    tmpl: Dict[str, int] = {}
    pre, suf, set_tmpl = [], [], []
    for k, key in enumerate(distinct):
        idxs = []
        for p in key:
            t = tmpl.get(p)
            if t is None:
                if type(p) is not str or p.count("{}") != 1 or p.count("{") != 1 or p.count("}") != 1:
                    return None
                a, b = p.split("{}")
                t = tmpl[p] = len(pre)
                pre.append(a)
                suf.append(b)
            idxs.append(t)
        distinct[key] = k
        set_tmpl.append(np.asarray(idxs, dtype=np.int32))
    n = len(names)
    offsets = None
    if len(set_tmpl) == 1:
        per = int(set_tmpl[0].size)
        counts = [per] * n
        tmpl_idx = np.tile(set_tmpl[0], n)
        name_idx = np.repeat(np.arange(n, dtype=np.int32), per)
        offsets = np.arange(n + 1, dtype=np.int64) * per      # (the engine's segment offsets, without a 1 000-element list round trip)
    else:
        req_set = [distinct[key] for key in keys]
        counts = [int(set_tmpl[k].size) for k in req_set]
        tmpl_idx = np.concatenate([set_tmpl[k] for k in req_set])
        name_idx = np.repeat(np.arange(n, dtype=np.int32), counts)
    if tmpl_idx.size == 0:
        return None

    def prompt(i):
        t = int(tmpl_idx[i])
        return pre[t] + names[int(name_idx[i])] + suf[t]

    packed_names = host_text.pack_strings(names)
    ids, lengths, fb, name_last = twin.encode_templated(pre, suf, packed_names, tmpl_idx, name_idx, want_name_last=True, narrow=True)
    if fb.any():                # rows outside the native library (non-ASCII, special-token syntax): the HF tokenizer
        rows = np.nonzero(fb)[0].tolist()
        enc = tokenizer([prompt(i) for i in rows], padding=False, truncation=True)["input_ids"]
        for i, r in zip(rows, enc):
            ids[i, :len(r)] = r
            lengths[i] = len(r)
    S = int(lengths.max())
    # as tokenize_lists does on every call: the longest row against the public tokenizer call
    # (a prompt string that has been compared once is not compared again: both tokenizers are deterministic functions of it)
    j = int(lengths.argmax())
    pj, lj = prompt(j), int(lengths[j])
    row_j = ids[j, :lj].copy()
    seen = twin.__dict__.setdefault("_verified_prompts", {})

    def probe_agrees() -> bool:
        if seen.get(pj) == row_j.tobytes():
            return True
        probe = tokenizer([pj], padding=True, truncation=True)
        want = probe["input_ids"][0]
        mask_len = int(sum(probe["attention_mask"][0]))
        if lj != mask_len or row_j.tolist() != list(want) or set(probe.keys()) != {"input_ids", "attention_mask"}:
            host_text.NativeClipBpe.disable(tokenizer)      # never trust a twin that disagreed once
            return False
        if len(seen) > 4096:
            seen.clear()
        seen[pj] = row_j.tobytes()
        return True

    if not defer_probe and not probe_agrees():
        return None
    ids = ids[:, :S]

    def walk() -> np.ndarray:        # the reference's subject search on every row (find_token_range, causal_trace.py:1046-1103)
        lk_ = finder_for(tokenizer).last_tokens(ids, names, name_idx, packed=packed_names)
        bad = np.nonzero((lk_ < 0) | (lk_ >= S))[0]
        if bad.size:
            j_ = int(bad[0])
            raise ValueError(f"lookup index {int(lk_[j_])} outside the padded prompt (S={S}) for prompt {prompt(j_)!r}")
        return lk_

    # The lookup token of ``prefix + name + suffix`` is the name's last token — unless the walk finds the name earlier in the
    # prompt, or something about the row is special.  With ``defer_probe`` the positions known from the construction are used
    # at once and the walk, like the tokenizer cross-check, runs behind the first launches (0.2 ms per 3 000 rows); any
    # difference makes ``verify`` say no and the engine starts over with the walk up front.
    if defer_probe and not fb.any() and int(name_last.min()) >= 0 and int(name_last.max()) < S:
        lk = name_last.astype(np.int64)

        def verify() -> bool:
            return probe_agrees() and bool(np.array_equal(walk(), lk))
    else:
        lk = walk()
        verify = probe_agrees if defer_probe else None
    return PromptChunk(np.ascontiguousarray(ids[:, :int(lk.max()) + 1]), lk, counts, n, verify=verify, offsets=offsets)


def iter_prompt_chunks(tokenizer, requests: Sequence[Dict], n_chunks: int, defer_probe: bool = False):
    """The request list in ``n_chunks`` contiguous slices, each tokenized, searched and truncated on its own, lazily: the
    caller builds a slice's prefix trie and launches its share of the encoder forward before asking for the next slice,
    so the GPU works on slice i while the host tokenizes slice i+1 (no helper thread: the launches are asynchronous)."""
    n = len(requests)
    n_chunks = max(1, min(n_chunks, n))
    first = requests[0]
    finder = finder_for(tokenizer)
    for i in range(n_chunks):
        lo, hi = (n * i) // n_chunks, (n * (i + 1)) // n_chunks
        if os.environ.get("EMCID_TEMPLATED", "1") != "0":
            fast = templated_prompt_chunk(tokenizer, requests[lo:hi], first, defer_probe)
            if fast is not None:
                yield fast
                continue
        prompts, subjects, counts = expand_request_prompts(requests[lo:hi], first)
        ids = tokenize_lists(tokenizer, prompts)["input_ids"]
        lookup = [r[-1] - 1 for r in finder.batch(ids, subjects)]
        lk = np.asarray(lookup, dtype=np.int64)
        bad = np.nonzero((lk < 0) | (lk >= ids.shape[1]))[0]
        if bad.size:
            j = int(bad[0])
            raise ValueError(f"lookup index {lookup[j]} outside the padded prompt (S={ids.shape[1]}) for prompt {prompts[j]!r}")
        if sum(counts) != len(prompts):
            raise ValueError(f"request prompt counts ({sum(counts)}) do not cover the {len(prompts)} prompts")
        yield PromptChunk(np.ascontiguousarray(ids[:, :int(lk.max()) + 1]), lookup, counts, hi - lo)


def gather_request_means(act: torch.Tensor, batch: PromptBatch) -> torch.Tensor:
    """(B, S, c) activations -> (N, c): row at each prompt's lookup token, averaged per request.  A multi-token batch
    (``lookup_multi``, k rows per prompt) gives (N k, c) in the reference's "rq num" order (emcid_main.py:993-1014)."""
    if act.dtype != torch.float32:
        raise hip.EmcidHipError(f"K/Z assembly is fp32 (reference loads the encoder in fp32); got {act.dtype}")
    if act.stride(-1) != 1:
        act = act.contiguous()
    if batch.lookup_multi is not None:
        per = [hip.gather_mean(act, col, batch.seg) for col in batch.lookup_multi]          # k x (N, c)
        return torch.stack(per, dim=1).reshape(-1, act.shape[-1])
    return hip.gather_mean(act, batch.lookup, batch.seg)


def build_prompt_batch_multi(tokenizer, requests: Sequence[Dict], device, k: int) -> PromptBatch:
    """The prompt batch of the ``num_fact_token = k > 1`` branch (reference: compute_z.py:2329-2360): every prompt padded to
    (longest + k - 2) tokens with padding="max_length"; per prompt the rows [last subject token, EOS, the k - 2 positions
    behind it].  No truncation: those rows lie at and behind the EOS."""
    if k < 2:
        raise ValueError(f"num_fact_token must be >= 2 here, got {k}")
    prompts, subjects, counts = expand_request_prompts(requests)
    first = tokenizer(prompts, padding=True, truncation=True)
    n_pad = k - 2
    enc = tokenizer(prompts, padding="max_length", truncation=True, max_length=len(first["input_ids"][0]) + n_pad)
    ids = np.asarray(enc["input_ids"], dtype=np.int64)
    mask = np.asarray(enc["attention_mask"], dtype=np.int64)
    last_subject = np.asarray([r[-1] - 1 for r in finder_for(tokenizer).batch(ids, subjects)], dtype=np.int64)
    eos = mask.sum(axis=1) - 1
    idx = np.concatenate([last_subject[:, None], eos[:, None] + np.arange(n_pad + 1)[None, :]], axis=1)      # (B, k)
    if idx.max() >= ids.shape[1] or idx.min() < 0:
        raise ValueError("lookup index outside the padded prompt")
    seg = np.cumsum([0] + counts).astype(np.int64)
    if seg[-1] != len(prompts):
        raise ValueError(f"request prompt counts ({seg[-1]}) do not cover the {len(prompts)} prompts")
    return PromptBatch(inputs={"input_ids": torch.from_numpy(ids).to(device), "attention_mask": torch.from_numpy(mask).to(device)},
                       lookup=torch.from_numpy(np.ascontiguousarray(idx[:, 0])).to(device), seg=torch.from_numpy(seg).to(device),
                       n_requests=len(requests), lookup_host=idx[:, 0].tolist(), ids_host=ids,
                       lookup_multi=torch.from_numpy(np.ascontiguousarray(idx.T)).to(device))


def get_module_input_output_at_words(text_encoder, tok, requests: List[Dict], module_name: str,
                                     num_fact_token: int = 1, batch: Optional[PromptBatch] = None
                                     ) -> Tuple[torch.Tensor, torch.Tensor]:
    """(input_ret (N, d), output_ret (N, h)) of ``module_name`` at the last subject token, mean over each
    request's prompts (reference: compute_z.py:2252-2325); with ``num_fact_token = k > 1`` the rows [last subject token,
    EOS, k - 2 padding positions] of prompts padded to (longest + k - 2) tokens: (N, k, d), (N, k, h) (:2329-2382)."""
    if num_fact_token != 1:
        return _module_input_output_multi(text_encoder, tok, requests, module_name, int(num_fact_token))
    device = next(text_encoder.parameters()).device
    if batch is None:
        batch = build_prompt_batch(tok, requests, device)
    grabbed = {}

    def hook(mod, inputs, output):
        grabbed["in"], grabbed["out"] = inputs[0], output
        raise StopForward()

    handle = get_module(text_encoder, module_name).register_forward_hook(hook)
    try:
        with torch.no_grad(), hip_attention(text_encoder):
            try:
                text_encoder(**batch.inputs)
            except StopForward:
                pass
    finally:
        handle.remove()
    return gather_request_means(grabbed["in"], batch), gather_request_means(grabbed["out"], batch)


def _module_input_output_multi(text_encoder, tok, requests: List[Dict], module_name: str, k: int):
    """The ``num_fact_token > 1`` branch (reference: compute_z.py:2329-2382): (N, k, d) inputs and (N, k, h) outputs of
    ``module_name`` at [last subject token, EOS, k - 2 padding positions], mean over each request's prompts."""
    if k < 2:
        raise ValueError(f"num_fact_token must be >= 1, got {k}")
    device = next(text_encoder.parameters()).device
    batch = build_prompt_batch_multi(tok, requests, device, k)
    grabbed = {}

    def hook(mod, inputs, output):
        grabbed["in"], grabbed["out"] = inputs[0], output
        raise StopForward()

    handle = get_module(text_encoder, module_name).register_forward_hook(hook)
    try:
        with torch.no_grad(), hip_attention(text_encoder):
            try:
                text_encoder(**batch.inputs)
            except StopForward:
                pass
    finally:
        handle.remove()
    n = len(requests)
    return (gather_request_means(grabbed["in"], batch).reshape(n, k, -1),          # one gather + per-request mean per looked-up
            gather_request_means(grabbed["out"], batch).reshape(n, k, -1))         # position (bit-compatible with torch's mean)


# ---- Stage 1: v* by Adam through the UNet (reference: emcid/compute_z.py:34-53, :315-649) ----------------------------------

def preprocess_img(images, resolution: int = 512) -> torch.Tensor:
    """Resize(bilinear) -> CenterCrop -> RandomHorizontalFlip -> ToTensor -> Normalize(0.5, 0.5) of PIL images, (n, 3, R, R)
    fp32 on the host (reference :34-53, whose torchvision transforms are restated on PIL + torch: torchvision is not a
    dependency here).  One ``torch.rand(1)`` per image decides the flip, as torchvision's RandomHorizontalFlip does."""
    from PIL import Image
    out = []
    for im in images:
        im = im.convert("RGB")
        w, h = im.size
        if not ((w <= h and w == resolution) or (h <= w and h == resolution)):
            im = im.resize((resolution, int(resolution * h / w)) if w < h else (int(resolution * w / h), resolution), Image.BILINEAR)
            w, h = im.size
        left, top = int(round((w - resolution) / 2.0)), int(round((h - resolution) / 2.0))
        im = im.crop((left, top, left + resolution, top + resolution))
        if torch.rand(1) < 0.5:
            im = im.transpose(Image.FLIP_LEFT_RIGHT)
        x = torch.from_numpy(np.asarray(im, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0)
        out.append((x - 0.5) / 0.5)
    return torch.stack(out)


def default_noise_scheduler():
    """The DDPM training schedule of Stable Diffusion v1.x (reference :378 loads it from the hub): diffusers' own class
    when that package and its files are reachable, else the same schedule from its published constants."""
    try:
        from diffusers import DDPMScheduler
        return DDPMScheduler.from_pretrained("CompVis/stable-diffusion-v1-4", subfolder="scheduler")
    except Exception:
        from .synthetic import DDPMNoiseSchedule
        return DDPMNoiseSchedule()


FIM_FILE = "data/fim_stats/text_encoder/ccs_filtered_stats/text_model.encoder.layers.10.mlp.fc2_float32_mean_step10_3000.npz"


def load_fim(device, path: Optional[str] = None) -> torch.Tensor:
    """The Fisher diagonal Stage 1's EWC term weighs ``delta ** 2`` with (reference: emcid/compute_z.py:478-486 — a ``Mean``
    statistic that emcid/fim_cal.py wrote, read from the same cwd-relative path; ``EMCID_FIM_FILE`` or ``path`` override it)."""
    from .runningstats import CombinedStat, Mean
    file_path = path or os.environ.get("EMCID_FIM_FILE") or FIM_FILE
    stat = CombinedStat(**{"mean": Mean()})
    with np.load(file_path, allow_pickle=True) as data:
        stat.load_state_dict(data)
    return torch.from_numpy(stat.mean.state_dict()["mean"]).to(device)


def compute_z_text_encoder(pipe, request: Dict, hparams, layer: int, device=None, noise_scheduler=None,
                           resolution: int = 512, rng_device=None) -> torch.Tensor:
    """v* of one concept: the hidden state of ``layer_module_tmp.format(layer)`` at the last subject token of the first
    prompt plus a vector ``delta`` found by ``v_num_grad_steps`` Adam steps on the denoising objective (MSE between the
    UNet's predictions under the edited source embedding and under the destination embedding, or the sampled noise),
    a weight decay and a text-alignment term, projected onto an L2 ball (reference: emcid/compute_z.py:315-649;
    same arguments, same return).  Runs wherever the pipeline lives (PyTorch-ROCm autograd through ``pipe.unet``).

    Results-identical restructuring: the text encoder is hooked in place instead of deep-copied per concept (it is frozen:
    both see the same weights); the clean text-encoder forwards, which the reference repeats every step, run once; the
    VAE's posterior of a training batch is computed once per distinct batch and SAMPLED every step like the reference's
    ``encode(...).latent_dist.sample()``.  Random draws happen in the reference's order (image flips, per step: sample
    indices, VAE posterior noise, latent noise, timesteps).  ``rng_device``: where they are drawn — default: like the
    reference on this device (the global generator of the model's device; sample indices and flips always on the host);
    "cpu": everything from the host generator, then moved (reproduces a CPU run on the GPU to fp32 rounding)."""
    from PIL import Image
    hp = hparams
    te = pipe.text_encoder
    dev = next(te.parameters()).device
    fim = load_fim(dev) if getattr(hp, "use_ewc", False) else None          # :478-486 (two shipped hparams files set use_ewc)
    rdev = torch.device(rng_device) if rng_device is not None else dev
    tok = pipe.tokenizer
    sched = noise_scheduler if noise_scheduler is not None else default_noise_scheduler()
    objective = hp.objective
    if objective not in ("ablate-source", "ablate-dest", "esd"):
        raise ValueError(f"Objective {objective} can not be used for compute_z.")
    source_prompts = [p.format(request["source"]) for p in request["prompts"]]
    dest_prompts = ["" for _ in request["prompts"]] if objective == "esd" else [p.format(request["dest"]) for p in request["prompts"]]
    spp = hp.samples_per_prompt
    if "training_img_paths" in request and objective != "esd":
        images = [Image.open(path) for path in request["training_img_paths"]]
    elif "images" in request and objective != "esd":
        images = request["images"]
    else:       # the reference samples the training images from the pipeline itself (:383-412)
        gen = torch.Generator(dev).manual_seed(int(request["seed_train"])) if request.get("seed_train") is not None else None
        images = []
        for _ in range(spp):
            images.extend(pipe(source_prompts, guidance_scale=7.5, generator=gen).images)
    pixels = preprocess_img(images, resolution)
    bsz = len(source_prompts)
    pixels = pixels.reshape(spp, bsz, *pixels.shape[1:]).transpose(0, 1)              # "(s b) c h w -> b s c h w"
    if len(pixels) % bsz:
        raise AssertionError(f"len(img_batch) {len(pixels)} should be n times of batch size {bsz}")
    src_inp, dst_inp = tokenize_prompts(source_prompts, tok, dev), tokenize_prompts(dest_prompts, tok, dev)
    finder = finder_for(tok)
    src_lookup = [finder(ids, request["source"])[-1] - 1 for ids in src_inp["input_ids"].tolist()]
    dst_lookup = [finder(ids, request["dest"])[-1] - 1 for ids in dst_inp["input_ids"].tolist()]
    if not (len(src_inp["input_ids"]) == len(dst_inp["input_ids"]) == len(pixels)):
        raise AssertionError("The number of prompts and images should be the same.")
    ar = torch.arange(bsz, device=dev)
    src_idx, dst_idx = torch.tensor(src_lookup, device=dev), torch.tensor(dst_lookup, device=dev)
    frozen = [prm for m in (te, pipe.vae, pipe.unet) for prm in m.parameters() if prm.requires_grad]
    for prm in frozen:
        prm.requires_grad_(False)
    delta = torch.zeros((te.config.hidden_size,), requires_grad=True, device=dev)
    opt = torch.optim.Adam([delta], lr=hp.v_lr)
    state = {"edit": False, "source_init": None}

    def hook(mod, args, out):
        if not state["edit"]:
            return out
        h = out[0] if isinstance(out, tuple) else out          # transformers 4.x layers return a tuple, 5.x the tensor
        if state["source_init"] is None:
            state["source_init"] = h[0, src_lookup[0]].detach().clone()
        h = h.clone()
        # one prompt after the other like the reference's hook (:353-373: cur_out[0][i, idx, :] += delta): autograd then sums
        # delta's gradient over the prompts in the reference's order — a vectorised add reduces them in another order, a
        # last-bit difference that 200 Adam steps amplify to 8e-5 of |v*| (tests/test_oracle_golden.py, toy_stage1_more)
        for i, idx in enumerate(src_lookup):
            if hp.replace_repr:
                h[i, idx, :] = delta
            else:
                h[i, idx, :] += delta
        return (h,) + tuple(out[1:]) if isinstance(out, tuple) else h

    def edited(inp):
        state["edit"] = True
        try:
            return te(**inp)[0:2]
        finally:
            state["edit"] = False

    handle = get_module(te, hp.layer_module_tmp.format(layer)).register_forward_hook(hook)
    try:
        with torch.no_grad():        # loop invariants (the reference recomputes them every step from the same frozen encoder)
            dest_repr, dest_pool = te(**dst_inp)[0:2]
            source_repr = te(**src_inp)[0] if (objective == "esd" or hp.cal_text_repr_loss) else None
            if hp.contrastive_text_loss:
                neg_pool = te(**tokenize_prompts(request["negative_prompts"], tok, dev))[1]
                single_pool = te(**tokenize_prompts([request["dest"]], tok, dev))[1]
            if hp.align_obj_eos_pad:
                full = lambda ps: {k: v.to(dev) for k, v in tok(ps, max_length=tok.model_max_length, return_tensors="pt",
                                                                padding="max_length", truncation=True).items()}
                src_full, dst_full = full(source_prompts), full(dest_prompts)
                src_eos = [int(m.sum()) - 1 for m in src_full["attention_mask"]]
                dst_eos = [int(m.sum()) - 1 for m in dst_full["attention_mask"]]
                far = max(src_eos + dst_eos)
                src_slices = [list(range(e, tok.model_max_length - max(0, far - e))) for e in src_eos]
                dst_slices = [list(range(e, tok.model_max_length - max(0, far - e))) for e in dst_eos]
                dest_full = te(**dst_full)[0]
                d_pad = torch.stack([dest_full[i, sl, :] for i, sl in enumerate(dst_slices)], dim=0)
        posteriors = {}
        host_draw = rdev.type == "cpu" and dev.type != "cpu"
        for it in range(hp.v_num_grad_steps):
            opt.zero_grad()
            sample_indices = torch.randint(0, spp, (bsz,))
            key = tuple(sample_indices.tolist())
            if key not in posteriors:
                with torch.no_grad():
                    posteriors[key] = pipe.vae.encode(pixels[torch.arange(bsz), sample_indices].to(dev)).latent_dist
            with torch.no_grad():
                latents = posteriors[key].sample(torch.default_generator) if host_draw else posteriors[key].sample()
                latents = latents * pipe.vae.config.scaling_factor
            if host_draw:
                noise = torch.randn(latents.shape, dtype=latents.dtype).to(dev)
                timesteps = torch.randint(0, sched.config.num_train_timesteps, (bsz,)).long().to(dev)
            else:
                noise = torch.randn_like(latents, device=dev)
                timesteps = torch.randint(0, sched.config.num_train_timesteps, (bsz,), device=dev).long()
            noisy = sched.add_noise(latents, noise, timesteps)
            edit_repr, edit_pool = edited(src_inp)
            source_init = state["source_init"]
            if not hp.no_noise_loss:
                edit_pred = pipe.unet(noisy, timesteps, edit_repr).sample
                with torch.no_grad():
                    pred_dest = pipe.unet(noisy, timesteps, dest_repr).sample
            if fim is not None and "ablate" in objective:        # EWC instead of the weight decay (:547-549; esd keeps the decay, :553)
                reg = torch.sum(float(hp.ewc_lambda) * fim * delta ** 2) / (2 * torch.norm(source_init) ** 2)
            else:
                reg = hp.v_weight_decay * (torch.norm(delta) / torch.norm(source_init) ** 2)
            if "ablate" in objective:
                if getattr(hp, "use_sampled_noise", False) or request.get("use_real_noise", False):
                    loss = F.mse_loss(noise, edit_pred, reduction="mean") + reg
                elif hp.no_noise_loss:
                    loss = reg
                else:
                    loss = F.mse_loss(edit_pred, pred_dest, reduction="mean") + reg
            else:
                with torch.no_grad():
                    pred_source = pipe.unet(noisy, timesteps, source_repr).sample
                loss = F.mse_loss(edit_pred, pred_dest - hp.esd_mu * (pred_source - pred_dest), reduction="mean") + reg
            if hp.cal_text_repr_loss and request.get("txt_align", True):
                scale = hp.text_repr_loss_scale_factor
                if hp.contrastive_text_loss:
                    emb = torch.cat([single_pool, neg_pool], dim=0)
                    scores = torch.squeeze(-torch.cdist(edit_pool.unsqueeze(0), emb.unsqueeze(0)))
                    loss = loss + scale * (-torch.log_softmax(scores, dim=1)[:, 0].mean(dim=0))
                elif hp.align_object_token:
                    loss = loss + scale * F.mse_loss(edit_repr[ar, src_idx, :], dest_repr[ar, dst_idx, :], reduction="mean")
                elif hp.align_obj_eos_pad:
                    e_full = edited(src_full)[0]
                    e_pad = torch.stack([e_full[i, sl, :] for i, sl in enumerate(src_slices)], dim=0)
                    loss = loss + scale * F.mse_loss(torch.cat([edit_repr[ar, src_idx, :].unsqueeze(1), e_pad], dim=1),
                                                     torch.cat([dest_repr[ar, dst_idx, :].unsqueeze(1), d_pad], dim=1),
                                                     reduction="mean")
                else:
                    loss = loss + scale * F.mse_loss(edit_pool, dest_pool, reduction="mean")
            loss.backward()
            opt.step()
            max_norm = hp.clamp_norm_factor * source_init.norm()
            if delta.norm() > max_norm:
                with torch.no_grad():
                    delta[...] = delta * max_norm / delta.norm()
    finally:
        handle.remove()
        for prm in frozen:
            prm.requires_grad_(True)
    return (state["source_init"] + delta).detach()


SLD_SAFE_WORDS = ("hate, harassment, violence, suffering, humiliation, harm, suicide, sexual, nudity, bodily fluids, blood, "
                  "obscene gestures, illegal activity, drug use, theft, vandalism, weapons, child abuse, brutality, cruelty")
SLD_PRESETS = {     # reference :2448-2466 (the "max" / "strong" configurations of safe latent diffusion)
    "max": dict(sld_guidance_scale=5000, sld_warmup_steps=0, sld_threshold=1.0, sld_momentum_scale=0.5, sld_mom_beta=0.7),
    "strong": dict(sld_guidance_scale=2000, sld_warmup_steps=7, sld_threshold=0.025, sld_momentum_scale=0.5, sld_mom_beta=0.7),
}


def compute_z_text_encoder_global(pipe, request: Dict, hparams, layer: int, device=None, noise_scheduler=None,
                                  resolution: int = 512, rng_device=None) -> torch.Tensor:
    """Stage 1 of a GLOBAL concept (``hparams.sld_supervision``; selected at reference emcid_main.py:911-918): one vector
    added to the output of ``layer_module_tmp.format(layer)`` at the start token (``request["source"] == "[CLS]"``) or the last
    position (``"[EOS]"``) of EVERY prompt of ``request["source_prompts"]``, optimised so that the UNet's prediction under the
    edited embedding follows the safe-latent-diffusion supervision — the clean prediction steered away from
    ``request["safe_words"]`` — or the esd form (reference: emcid/compute_z.py:77-312; same arguments, same return: the mean
    initial state at that position + delta).

    Results-identical restructuring as in ``compute_z_text_encoder``: the frozen encoder is hooked in place instead of deep-copied,
    the clean encoder forwards run once, no loss log file.  Random draws in the reference's order (image flips, the VAE
    posterior's noise ONCE before the loop — this variant samples its latents once, :203-205 —, per step latent noise and
    timesteps).  Training images: ``request["training_img_paths"]`` / ``request["images"]``, or one image per prompt sampled from
    the pipeline with its seed (``request["seeds"]``, :140-146 / :159-164); the reference's ``sld_generate`` (a second,
    safe-latent-diffusion pipeline fetched from the hub, :155) is outside this repository: ablate-dest needs the images given."""
    from PIL import Image
    hp = hparams
    te = pipe.text_encoder
    dev = next(te.parameters()).device
    rdev = torch.device(rng_device) if rng_device is not None else dev
    host_draw = rdev.type == "cpu" and dev.type != "cpu"
    tok = pipe.tokenizer
    sched = noise_scheduler if noise_scheduler is not None else default_noise_scheduler()
    objective = hp.objective
    if objective not in ("ablate-source", "ablate-dest", "esd"):
        raise ValueError(f"Objective {objective} can not be used for compute_z.")
    source_prompts = list(request["source_prompts"])
    if request["source"] == "[CLS]":
        edit_idx = 0
    elif request["source"] == "[EOS]":
        edit_idx = -1
    else:       # the reference leaves edit_idx unbound for any other source and stops at a NameError inside the hook (:108-111, :123)
        raise NameError("name 'edit_idx' is not defined: compute_z_text_encoder_global edits '[CLS]' or '[EOS]' only")
    if hp.sld_type not in SLD_PRESETS:
        raise ValueError(f"sld_type {hp.sld_type} not supported")
    sld = {k: torch.tensor(v).to(dev) for k, v in SLD_PRESETS[hp.sld_type].items()}
    if objective != "esd" and "training_img_paths" in request:
        images = [Image.open(path) for path in request["training_img_paths"]]
    elif objective != "esd" and "images" in request:
        images = request["images"]
    elif objective == "ablate-dest":
        raise NotImplementedError("ablate-dest without training images samples them with the reference's sld_generate "
                                  "(StableDiffusionPipelineSafe from the hub, compute_z.py:155): pass request['images'] or "
                                  "request['training_img_paths']")
    else:
        images = []
        for prompt, seed in zip(source_prompts, request["seeds"]):
            gen = torch.Generator(rdev if host_draw else dev).manual_seed(int(seed)) if seed is not None else None
            images.append(pipe([prompt], guidance_scale=7.5, generator=gen).images[0])
    pixels = preprocess_img(images, resolution).to(dev)
    src_inp = tokenize_prompts(source_prompts, tok, dev)
    safe_inp = tokenize_prompts(request["safe_words"], tok, dev)
    uncond_inp = tokenize_prompts([""] * pixels.shape[0], tok, dev)
    if len(src_inp["input_ids"]) != len(pixels):
        raise AssertionError("The number of prompts and images should be the same.")
    bsz = len(pixels)
    frozen = [prm for m in (te, pipe.vae, pipe.unet) for prm in m.parameters() if prm.requires_grad]
    for prm in frozen:
        prm.requires_grad_(False)
    delta = torch.zeros((te.config.hidden_size,), requires_grad=True, device=dev)
    opt = torch.optim.Adam([delta], lr=hp.v_lr)
    state = {"edit": False, "source_init": None}

    def hook(mod, args, out):
        if not state["edit"]:
            return out
        h = out[0] if isinstance(out, tuple) else out
        if state["source_init"] is None:
            state["source_init"] = h[:, edit_idx].detach().clone().mean(dim=0)
        h = h.clone()
        for i in range(bsz):          # prompt by prompt, like the reference's hook (:126-127): autograd sums in that order
            h[i, edit_idx, :] += delta
        return (h,) + tuple(out[1:]) if isinstance(out, tuple) else h

    handle = get_module(te, hp.layer_module_tmp.format(layer)).register_forward_hook(hook)
    try:
        with torch.no_grad():
            posterior = pipe.vae.encode(pixels).latent_dist
            latents = posterior.sample(torch.default_generator) if host_draw else posterior.sample()
            latents = latents * pipe.vae.config.scaling_factor
            safety_repr = te(**safe_inp)[0]
            source_repr = te(**src_inp)[0]
            uncond_repr = te(**uncond_inp)[0]
        for it in range(hp.v_num_grad_steps):
            opt.zero_grad()
            if host_draw:
                noise = torch.randn(latents.shape, dtype=latents.dtype).to(dev)
                timesteps = torch.randint(0, sched.config.num_train_timesteps, (bsz,)).long().to(dev)
            else:
                noise = torch.randn_like(latents, device=dev)
                timesteps = torch.randint(0, sched.config.num_train_timesteps, (bsz,), device=dev).long()
            noisy = sched.add_noise(latents, noise, timesteps)
            state["edit"] = True
            try:
                edit_repr = te(**src_inp)[0]
            finally:
                state["edit"] = False
            source_init = state["source_init"]
            with torch.no_grad():
                pred_source = pipe.unet(noisy, timesteps, source_repr).sample
                pred_uncond = pipe.unet(noisy, timesteps, uncond_repr).sample
                guidance = None
                if hp.sld_supervision:      # StableDiffusionPipelineSafe's guidance (:232-248)
                    pred_safety = pipe.unet(noisy, timesteps, safety_repr).sample
                    scale = torch.clamp(torch.abs(pred_source - pred_safety) * sld["sld_guidance_scale"], max=1.0)
                    concept_scale = torch.where((pred_source - pred_safety) >= sld["sld_threshold"], torch.zeros_like(scale), scale)
                    guidance = torch.mul(pred_safety - pred_uncond, concept_scale)
            edit_pred = pipe.unet(noisy, timesteps, edit_repr).sample
            decay = hp.v_weight_decay * (torch.norm(delta) / torch.norm(source_init) ** 2)
            if "ablate" in objective:
                if getattr(hp, "use_sampled_noise", False):
                    mse = F.mse_loss(noise, edit_pred, reduction="mean")
                else:
                    if guidance is None:      # the reference's NameError (:261: noise_guidance_safety is set under sld_supervision only)
                        raise NameError("name 'noise_guidance_safety' is not defined: compute_z_text_encoder_global needs "
                                        "hparams.sld_supervision (or use_sampled_noise / the esd objective)")
                    mse = F.mse_loss(edit_pred, pred_source - guidance, reduction="mean")
            else:
                mse = F.mse_loss(edit_pred, pred_uncond - hp.esd_mu * (pred_source - pred_uncond), reduction="mean")
            loss = mse + decay
            loss.backward()
            opt.step()
            max_norm = hp.clamp_norm_factor * source_init.norm()
            if delta.norm() > max_norm:
                with torch.no_grad():
                    delta[...] = delta * max_norm / delta.norm()
    finally:
        handle.remove()
        for prm in frozen:
            prm.requires_grad_(True)
    return (state["source_init"] + delta).detach()


CLIP_HUB_ID = "openai/clip-vit-large-patch14"      # reference compute_z.py:1376


def default_clip_towers():
    """(text tower with projection, vision tower with projection, processor) of the hub checkpoint the reference's
    compute_z_text_encoder_v1 loads (compute_z.py:1376-1378, :1440): needs the files (no network here -> OSError)."""
    from transformers import CLIPProcessor, CLIPTextModelWithProjection, CLIPVisionModelWithProjection
    return (CLIPTextModelWithProjection.from_pretrained(CLIP_HUB_ID), CLIPVisionModelWithProjection.from_pretrained(CLIP_HUB_ID),
            CLIPProcessor.from_pretrained(CLIP_HUB_ID))


def compute_z_text_encoder_v1(pipe, request: Dict, hparams, layer: int, device=None, noise_scheduler=None,
                              resolution: int = 512, rng_device=None, clip_towers=None) -> torch.Tensor:
    """The ``txt_img_align_scale_factor != 0`` Stage 1 (reference: emcid/compute_z.py:1360-1648; selected at
    emcid_main.py:919-926): as ``compute_z_text_encoder``, but the edited encoder is CLIP's text tower WITH its projection
    (``clip_towers[0]``; the reference loads openai/clip-vit-large-patch14, whose text tower is SD-v1.x's encoder), the
    text-alignment terms live in the projected space (``text_embeds`` against the projected pooled output of the destination
    prompts), and — ``request["txt_img_align"]`` — the projected text embedding is pulled towards the CLIP image embedding of the
    training images (``clip_towers[1]`` through ``clip_towers[2]``; "cos" or "l2", ablate-dest only as in the reference, where
    any other objective stops at a NameError on ``dest_img_emb``).  The latents are sampled ONCE before the loop (:1483-1485);
    one image per prompt (no ``samples_per_prompt``).  ``clip_towers``: default ``default_clip_towers()`` (hub files).

    Restructured like the other forms (results identical): hooked in place — the tower is the caller's, restored afterwards —,
    clean forwards hoisted, no loss log; random draws in the reference's order (pipeline sampling by ``seed_train``, image
    flips, VAE posterior noise once, per step latent noise and timesteps)."""
    from PIL import Image
    hp = hparams
    te = pipe.text_encoder
    dev = next(te.parameters()).device
    rdev = torch.device(rng_device) if rng_device is not None else dev
    host_draw = rdev.type == "cpu" and dev.type != "cpu"
    tok = pipe.tokenizer
    sched = noise_scheduler if noise_scheduler is not None else default_noise_scheduler()
    text_proj, vision_proj, processor = clip_towers if clip_towers is not None else default_clip_towers()
    text_proj = text_proj.to(dev)
    objective = hp.objective
    if objective not in ("ablate-source", "ablate-dest", "esd"):
        raise ValueError(f"Objective {objective} can not be used for compute_z.")
    align_img = bool(request["txt_img_align"])                     # (KeyError without the field, like the reference :1436)
    source_prompts = [p.format(request["source"]) for p in request["prompts"]]
    dest_prompts = ["" for _ in request["prompts"]] if objective == "esd" else [p.format(request["dest"]) for p in request["prompts"]]
    if objective != "esd" and "training_img_paths" in request:
        images = [Image.open(path) for path in request["training_img_paths"]]
    elif objective != "esd" and "images" in request:
        images = request["images"]
    else:
        gen = torch.Generator(rdev if host_draw else dev).manual_seed(int(request["seed_train"])) if request["seed_train"] is not None else None
        images = pipe(dest_prompts if objective == "ablate-dest" else source_prompts, guidance_scale=7.5, generator=gen).images
    dest_img_emb = None
    if objective == "ablate-dest" and align_img:
        with torch.no_grad():
            vision_proj = vision_proj.to(dev)
            img_inp = processor(images=images, return_tensors="pt").to(dev)
            dest_img_emb = vision_proj(**img_inp).image_embeds
    pixels = preprocess_img(images, resolution).to(dev)
    src_inp, dst_inp = tokenize_prompts(source_prompts, tok, dev), tokenize_prompts(dest_prompts, tok, dev)
    finder = finder_for(tok)
    src_lookup = [finder(ids, request["source"])[-1] - 1 for ids in src_inp["input_ids"].tolist()]
    dst_lookup = [finder(ids, request["dest"])[-1] - 1 for ids in dst_inp["input_ids"].tolist()]
    if not (len(src_inp["input_ids"]) == len(dst_inp["input_ids"]) == len(pixels)):
        raise AssertionError("The number of prompts and images should be the same.")
    bsz = len(pixels)
    ar = torch.arange(bsz, device=dev)
    src_idx, dst_idx = torch.tensor(src_lookup, device=dev), torch.tensor(dst_lookup, device=dev)
    fim = load_fim(dev) if getattr(hp, "use_ewc", False) else None
    frozen = [prm for m in (text_proj, te, pipe.vae, pipe.unet) for prm in m.parameters() if prm.requires_grad]
    for prm in frozen:
        prm.requires_grad_(False)
    delta = torch.zeros((text_proj.config.hidden_size,), requires_grad=True, device=dev)
    opt = torch.optim.Adam([delta], lr=hp.v_lr)
    state = {"source_init": None}

    def hook(mod, args, out):
        h = out[0] if isinstance(out, tuple) else out
        if state["source_init"] is None:
            state["source_init"] = h[0, src_lookup[0]].detach().clone()
        h = h.clone()
        for i, idx in enumerate(src_lookup):          # prompt by prompt (:1412-1417): autograd sums delta's gradient in that order
            if hp.replace_repr:
                h[i, idx, :] = delta
            else:
                h[i, idx, :] += delta
        return (h,) + tuple(out[1:]) if isinstance(out, tuple) else h

    handle = get_module(text_proj, hp.layer_module_tmp.format(layer)).register_forward_hook(hook)
    try:
        with torch.no_grad():
            posterior = pipe.vae.encode(pixels).latent_dist
            latents = posterior.sample(torch.default_generator) if host_draw else posterior.sample()
            latents = latents * pipe.vae.config.scaling_factor
            out_d = te(**dst_inp)
            dest_repr, dest_pool = out_d[0], out_d[1]
            dest_emb = text_proj.text_projection(dest_pool)
            source_repr = te(**src_inp)[0] if (objective == "esd" or hp.cal_text_repr_loss) else None
            if hp.cal_text_repr_loss and hp.contrastive_text_loss:
                neg_emb = text_proj.text_projection(te(**tokenize_prompts(request["negative_prompts"], tok, dev))[1])
                single_emb = text_proj.text_projection(te(**tokenize_prompts([request["dest"]], tok, dev))[1])
        for it in range(hp.v_num_grad_steps):
            opt.zero_grad()
            if host_draw:
                noise = torch.randn(latents.shape, dtype=latents.dtype).to(dev)
                timesteps = torch.randint(0, sched.config.num_train_timesteps, (bsz,)).long().to(dev)
            else:
                noise = torch.randn_like(latents, device=dev)
                timesteps = torch.randint(0, sched.config.num_train_timesteps, (bsz,), device=dev).long()
            noisy = sched.add_noise(latents, noise, timesteps)
            out_e = text_proj(**src_inp)
            edit_repr, edit_emb = out_e.last_hidden_state, out_e.text_embeds
            source_init = state["source_init"]
            edit_pred = pipe.unet(noisy, timesteps, edit_repr).sample
            with torch.no_grad():
                pred_dest = pipe.unet(noisy, timesteps, dest_repr).sample
            decay = hp.v_weight_decay * (torch.norm(delta) / torch.norm(source_init) ** 2)
            if "ablate" in objective:
                mse = F.mse_loss(noise, edit_pred, reduction="mean") if getattr(hp, "use_sampled_noise", False) \
                    else F.mse_loss(edit_pred, pred_dest, reduction="mean")
                reg = float(hp.ewc_lambda) * torch.sum(fim * delta ** 2) / (2 * torch.norm(source_init) ** 2) if fim is not None else decay
                loss = mse + reg
            else:
                with torch.no_grad():
                    pred_source = pipe.unet(noisy, timesteps, source_repr).sample
                loss = F.mse_loss(edit_pred, pred_dest - hp.esd_mu * (pred_source - pred_dest), reduction="mean") + decay
            if hp.cal_text_repr_loss and objective != "esd":
                scale = hp.text_repr_loss_scale_factor
                if hp.contrastive_text_loss:
                    emb = torch.cat([single_emb, neg_emb], dim=0)
                    scores = torch.squeeze(-torch.cdist(edit_emb.unsqueeze(0), emb.unsqueeze(0)))
                    loss = loss + scale * (-torch.log_softmax(scores, dim=1)[:, 0].mean(dim=0))
                elif hp.align_object_token:
                    loss = loss + scale * F.mse_loss(edit_repr[ar, src_idx, :], dest_repr[ar, dst_idx, :], reduction="mean")
                else:
                    loss = loss + scale * F.mse_loss(edit_emb, dest_emb, reduction="mean")
            if align_img:
                if dest_img_emb is None:          # the reference's NameError (:1441 defines it for ablate-dest only)
                    raise NameError("name 'dest_img_emb' is not defined: txt_img_align needs the ablate-dest objective")
                if hp.txt_img_align_loss_metric == "cos":
                    align = -(F.cosine_similarity(edit_emb, dest_img_emb, dim=1).mean() - 1)
                elif hp.txt_img_align_loss_metric == "l2":
                    align = F.mse_loss(edit_emb, dest_img_emb, reduction="mean")
                else:
                    raise ValueError(f"txt_img_align_loss_metric {hp.txt_img_align_loss_metric} not supported")
                loss = loss + hp.txt_img_align_scale_factor * align
            loss.backward()
            opt.step()
            max_norm = hp.clamp_norm_factor * source_init.norm()
            if delta.norm() > max_norm:
                with torch.no_grad():
                    delta[...] = delta * max_norm / delta.norm()
    finally:
        handle.remove()
        for prm in frozen:
            prm.requires_grad_(True)
    return (state["source_init"] + delta).detach()


def compute_z_unet_x_kv(pipe, request: Dict, hparams, device=None, noise_scheduler=None, resolution: int = 512,
                        rng_device=None) -> Dict[str, torch.Tensor]:
    """Stage 1 of the cross-attention sibling: the target vector of EVERY ``attn2.to_k`` / ``to_v`` projection of the UNet (16
    blocks x 2 for SD-v1.x) for one concept — the projection's output at the last subject token of the first prompt plus a
    delta found by ONE Adam over all of them on the denoising objective: MSE between the edited UNet's prediction and a
    supervision built from the clean UNet's predictions (safe-latent-diffusion guidance away from the ``safe words``, or the esd
    form), plus the mean weight decay, each delta projected onto its own L2 ball (reference: emcid/compute_z.py:2407-2645; same
    arguments, same return: {layer name: (out_features,) tensor}; called on a v* miss at emcid_main.py:398).

    Results-identical restructuring as in ``compute_z_text_encoder``: the UNet is hooked in place instead of deep-copied (its
    clean passes run with the hook switched off), the hook adds delta prompt by prompt like the reference's edit_output_fn
    (:2482-2492) so that autograd sums delta's gradient in the reference's order; no loss log file.  Random draws in the
    reference's order (image flips; per step sample indices, VAE posterior noise, latent noise, timesteps).  The reference
    defines ``samples_per_prompt`` only when it samples the training images from the pipeline (:2503-2510; with
    ``training_img_paths`` it stops at a NameError): here hparams.samples_per_prompt in every case, and ``request["images"]``
    is accepted like in the other Stage-1 forms."""
    from PIL import Image
    hp = hparams
    te, unet = pipe.text_encoder, pipe.unet
    dev = next(unet.parameters()).device
    rdev = torch.device(rng_device) if rng_device is not None else dev
    host_draw = rdev.type == "cpu" and dev.type != "cpu"
    tok = pipe.tokenizer
    sched = noise_scheduler if noise_scheduler is not None else default_noise_scheduler()
    if not hp.sld_supervision and hp.objective != "esd":
        raise ValueError("compute_z_unet_x_kv needs sld_supervision or objective == 'esd' (reference :2566-2590 defines no "
                         "supervision otherwise)")
    source_prompts = [p.format(request["source"]) for p in request["prompts"]]
    src_inp = tokenize_prompts(source_prompts, tok, dev)
    finder = finder_for(tok)
    src_lookup = [finder(ids, request["source"])[-1] - 1 for ids in src_inp["input_ids"].tolist()]
    bsz = len(source_prompts)
    sld = None
    if hp.sld_supervision:
        safe_words = SLD_SAFE_WORDS if hp.all_safe else request["safe words"]
        if hp.sld_type not in SLD_PRESETS:
            raise ValueError(f"sld_type {hp.sld_type} not supported")
        sld = {k: torch.tensor(v).to(dev) for k, v in SLD_PRESETS[hp.sld_type].items()}
    from .layer_stats import get_all_cross_attn_kv_layer_names
    names = get_all_cross_attn_kv_layer_names(pipe)            # the reference's order (down, up, mid; to_k then to_v per block)
    mods = dict(unet.named_modules())
    frozen = [prm for m in (unet, pipe.vae, te) for prm in m.parameters() if prm.requires_grad]
    for prm in frozen:
        prm.requires_grad_(False)
    deltas = {n: torch.zeros((mods[n].out_features,), requires_grad=True, device=dev) for n in names}
    inits: Dict[str, Optional[torch.Tensor]] = {n: None for n in names}
    state = {"edit": False}

    def make_hook(name):
        def hook(mod, args, out):
            if not state["edit"]:
                return out
            if inits[name] is None:
                inits[name] = out[0, src_lookup[0]].detach().clone()
            out = out.clone()
            for i, idx in enumerate(src_lookup):
                if hp.replace_repr:
                    out[i, idx, :] = deltas[name]
                else:
                    out[i, idx, :] += deltas[name]
            return out
        return hook

    opt = torch.optim.Adam([deltas[n] for n in names], lr=hp.v_lr)
    spp = hp.samples_per_prompt
    if "training_img_paths" in request:
        images = [Image.open(path) for path in request["training_img_paths"]]
    elif "images" in request:
        images = request["images"]
    else:
        gen = torch.Generator(dev).manual_seed(int(request["seed_train"]))
        images = []
        for _ in range(spp):
            images.extend(pipe(source_prompts, guidance_scale=7.5, generator=gen).images)
    if len(images) % bsz:
        raise AssertionError(f"len(img_batch) {len(images)} should be n times of batch size {bsz}")
    pixels = preprocess_img(images, resolution)
    pixels = pixels.reshape(spp, bsz, *pixels.shape[1:]).transpose(0, 1)              # "(s b) c h w -> b s c h w"
    handles = [mods[n].register_forward_hook(make_hook(n)) for n in names]
    try:
        with torch.no_grad():
            source_repr = te(**src_inp)[0]
            safe_repr = te(**tokenize_prompts([safe_words] * bsz, tok, dev))[0] if sld is not None else None
            uncond_repr = te(**tokenize_prompts([""] * bsz, tok, dev))[0]
        posteriors = {}
        for it in range(hp.v_num_grad_steps):
            opt.zero_grad()
            sample_indices = torch.randint(0, spp, (bsz,))
            key = tuple(sample_indices.tolist())
            if key not in posteriors:
                with torch.no_grad():
                    posteriors[key] = pipe.vae.encode(pixels[torch.arange(bsz), sample_indices].to(dev)).latent_dist
            with torch.no_grad():
                latents = posteriors[key].sample(torch.default_generator) if host_draw else posteriors[key].sample()
                latents = latents * pipe.vae.config.scaling_factor
            if host_draw:
                noise = torch.randn(latents.shape, dtype=latents.dtype).to(dev)
                timesteps = torch.randint(0, sched.config.num_train_timesteps, (bsz,)).long().to(dev)
            else:
                noise = torch.randn_like(latents, device=dev)
                timesteps = torch.randint(0, sched.config.num_train_timesteps, (bsz,), device=dev).long()
            noisy = sched.add_noise(latents, noise, timesteps)
            with torch.no_grad():
                pred_source = unet(noisy, timesteps, source_repr).sample
                pred_uncond = unet(noisy, timesteps, uncond_repr).sample
                if sld is not None:      # the safe-latent-diffusion guidance of StableDiffusionPipelineSafe (:2566-2584)
                    pred_safety = unet(noisy, timesteps, safe_repr).sample
                    scale = torch.clamp(torch.abs(pred_source - pred_safety) * sld["sld_guidance_scale"], max=1.0)
                    concept_scale = torch.where((pred_source - pred_safety) >= sld["sld_threshold"], torch.zeros_like(scale), scale)
                    supervision = pred_source - torch.mul(pred_safety - pred_uncond, concept_scale)
                else:
                    supervision = pred_uncond - hp.esd_mu * (pred_source - pred_uncond)
            state["edit"] = True
            try:
                edit_pred = unet(noisy, timesteps, source_repr).sample
            finally:
                state["edit"] = False
            mse = F.mse_loss(edit_pred, supervision, reduction="mean")
            decay = 0
            for n in names:
                decay += hp.v_weight_decay * (torch.norm(deltas[n]) / torch.norm(inits[n]) ** 2)
            loss = mse + decay / len(names)
            loss.backward()
            opt.step()
            for n in names:
                max_norm = hp.clamp_norm_factor * inits[n].norm()
                if deltas[n].norm() > max_norm:
                    with torch.no_grad():
                        deltas[n][...] = deltas[n] * max_norm / deltas[n].norm()
    finally:
        for hd in handles:
            hd.remove()
        for prm in frozen:
            prm.requires_grad_(True)
    with torch.no_grad():
        return {n: inits[n] + deltas[n] for n in names}


def compute_z_text_encoder_v2(pipe, request: Dict, hparams, layer: int, device=None, noise_scheduler=None,
                              resolution: int = 512, rng_device=None) -> torch.Tensor:
    """The ``use_new_compute_z`` Stage 1: ``hparams.num_edit_tokens`` vectors per concept, (k, hidden) — row 0 for the last
    subject token, rows 1.. for the EOS token and the k - 2 padding positions behind it, with the prompts tokenized to
    (longest + k - 2) by padding="max_length" (reference: emcid/compute_z.py:1041-1357; same arguments, same return; the
    dispatch is emcid_main.py:927-936).  Same restructuring and random-draw order as ``compute_z_text_encoder``.

    What the reference's text does here, followed as is: for the ablate objectives the weight decay is built under no_grad
    from the row norms (:1277-1281) — a constant of the loss, so it is not formed at all; the text term is the MSE over the
    k looked-up rows of the edited source and the destination embeddings for k >= 2, the pooled outputs' MSE for k = 1
    (:1297-1317); the L2 ball is per row, against that row's own initial norm (:1339-1343).  ``objective == "esd"`` and
    ``use_ewc`` read ``source_init`` before it is ever assigned in the reference (:1275, :1292: UnboundLocalError on the first
    step); here they raise NotImplementedError up front."""
    from PIL import Image
    hp = hparams
    objective = hp.objective
    if objective not in ("ablate-source", "ablate-dest", "esd"):
        raise ValueError(f"Objective {objective} can not be used for compute_z.")
    if objective == "esd" or getattr(hp, "use_ewc", False):
        raise NotImplementedError("compute_z_text_encoder_v2: the reference fails on esd / use_ewc (source_init read before assignment, "
                                  "compute_z.py:1275, :1292)")
    k = int(hp.num_edit_tokens)
    if k < 1:
        raise ValueError(f"num_edit_tokens must be >= 1, got {k}")
    te = pipe.text_encoder
    dev = next(te.parameters()).device
    rdev = torch.device(rng_device) if rng_device is not None else dev
    tok = pipe.tokenizer
    sched = noise_scheduler if noise_scheduler is not None else default_noise_scheduler()
    source_prompts = [p.format(request["source"]) for p in request["prompts"]]
    dest_prompts = [p.format(request["dest"]) for p in request["prompts"]]
    spp = hp.samples_per_prompt
    if "training_img_paths" in request:
        images = [Image.open(path) for path in request["training_img_paths"]]
    elif "images" in request:
        images = request["images"]
    else:
        gen = torch.Generator(dev).manual_seed(int(request["seed_train"])) if request.get("seed_train") is not None else None
        images = []
        for _ in range(spp):
            images.extend(pipe(source_prompts, guidance_scale=7.5, generator=gen).images)
    pixels = preprocess_img(images, resolution)
    bsz = len(source_prompts)
    pixels = pixels.reshape(spp, bsz, *pixels.shape[1:]).transpose(0, 1)              # "(s b) c h w -> b s c h w"
    if len(pixels) % bsz:
        raise AssertionError(f"len(img_batch) {len(pixels)} should be n times of batch size {bsz}")
    src_inp, dst_inp = tokenize_prompts(source_prompts, tok, dev), tokenize_prompts(dest_prompts, tok, dev)
    n_pad = k - 2
    if k > 1:
        padded = max(src_inp["input_ids"].shape[1], dst_inp["input_ids"].shape[1]) + n_pad
        src_inp = tokenize_prompts(source_prompts, tok, dev, padding_length=padded)
        dst_inp = tokenize_prompts(dest_prompts, tok, dev, padding_length=padded)
    finder = finder_for(tok)

    def lookup_rows(inp, subject):       # (B, k): [last subject token, EOS, EOS + 1, ...]
        rows = [[finder(ids, subject)[-1] - 1] for ids in inp["input_ids"].tolist()]
        if k >= 2:
            eos = (inp["attention_mask"].sum(dim=1) - 1).tolist()
            rows = [r + list(range(e, e + n_pad + 1)) for r, e in zip(rows, eos)]
        t = torch.tensor(rows, device=dev)
        if int(t.max()) >= inp["input_ids"].shape[1] or int(t.min()) < 0:
            raise ValueError("lookup index outside the padded prompt")
        return t

    src_idx, dst_idx = lookup_rows(src_inp, request["source"]), lookup_rows(dst_inp, request["dest"])
    src_rows = src_idx.tolist()
    if not (len(src_inp["input_ids"]) == len(dst_inp["input_ids"]) == len(pixels)):
        raise AssertionError("The number of prompts and images should be the same.")
    ar = torch.arange(bsz, device=dev)[:, None]
    frozen = [prm for m in (te, pipe.vae, pipe.unet) for prm in m.parameters() if prm.requires_grad]
    for prm in frozen:
        prm.requires_grad_(False)
    deltas = torch.zeros((k, te.config.hidden_size), requires_grad=True, device=dev)
    opt = torch.optim.Adam([deltas], lr=hp.v_lr)
    state = {"edit": False, "inits": None}

    def hook(mod, args, out):
        if not state["edit"]:
            return out
        h = out[0] if isinstance(out, tuple) else out
        if state["inits"] is None:
            state["inits"] = h[0, src_idx[0]].detach().clone()           # (k, hidden): the rows of the FIRST prompt
        h = h.clone()
        for i in range(bsz):                 # prompt by prompt, row by row: the reference's order (:1135-1141), see compute_z_text_encoder
            for j in range(k):
                if hp.replace_repr:
                    h[i, src_rows[i][j], :] = deltas[j, :]
                else:
                    h[i, src_rows[i][j], :] += deltas[j, :]
        return (h,) + tuple(out[1:]) if isinstance(out, tuple) else h

    def edited(inp):
        state["edit"] = True
        try:
            return te(**inp)[0:2]
        finally:
            state["edit"] = False

    handle = get_module(te, hp.layer_module_tmp.format(layer)).register_forward_hook(hook)
    try:
        with torch.no_grad():
            dest_repr, dest_pool = te(**dst_inp)[0:2]
            wanted_rows = dest_repr[ar, dst_idx, :]
        posteriors = {}
        host_draw = rdev.type == "cpu" and dev.type != "cpu"
        for it in range(hp.v_num_grad_steps):
            opt.zero_grad()
            sample_indices = torch.randint(0, spp, (bsz,))
            key = tuple(sample_indices.tolist())
            if key not in posteriors:
                with torch.no_grad():
                    posteriors[key] = pipe.vae.encode(pixels[torch.arange(bsz), sample_indices].to(dev)).latent_dist
            with torch.no_grad():
                latents = posteriors[key].sample(torch.default_generator) if host_draw else posteriors[key].sample()
                latents = latents * pipe.vae.config.scaling_factor
            if host_draw:
                noise = torch.randn(latents.shape, dtype=latents.dtype).to(dev)
                timesteps = torch.randint(0, sched.config.num_train_timesteps, (bsz,)).long().to(dev)
            else:
                noise = torch.randn_like(latents, device=dev)
                timesteps = torch.randint(0, sched.config.num_train_timesteps, (bsz,), device=dev).long()
            noisy = sched.add_noise(latents, noise, timesteps)
            edit_repr, edit_pool = edited(src_inp)
            loss = None
            if not hp.no_noise_loss:
                edit_pred = pipe.unet(noisy, timesteps, edit_repr).sample
                if getattr(hp, "use_sampled_noise", False) or request.get("use_real_noise", False):
                    loss = F.mse_loss(noise, edit_pred, reduction="mean")
                else:
                    with torch.no_grad():
                        pred_dest = pipe.unet(noisy, timesteps, dest_repr).sample
                    loss = F.mse_loss(edit_pred, pred_dest, reduction="mean")
            if hp.cal_text_repr_loss and request.get("txt_align", True):
                if k >= 2:
                    term = F.mse_loss(edit_repr[ar, src_idx, :], wanted_rows, reduction="mean")
                else:
                    term = F.mse_loss(edit_pool, dest_pool, reduction="mean")
                loss = hp.text_repr_loss_scale_factor * term if loss is None else loss + hp.text_repr_loss_scale_factor * term
            if loss is not None:         # (no_noise_loss without a text term: the loss is the constant weight decay, no gradient)
                loss.backward()
                opt.step()
            with torch.no_grad():
                max_norm = hp.clamp_norm_factor * state["inits"].norm(dim=1)
                norms = deltas.norm(dim=1)
                over = norms > max_norm
                if bool(over.any()):
                    scaled = deltas * max_norm.unsqueeze(1) / norms.unsqueeze(1)
                    deltas.copy_(torch.where(over.unsqueeze(1), scaled, deltas))
    finally:
        handle.remove()
        for prm in frozen:
            prm.requires_grad_(True)
    return (deltas + state["inits"]).detach()


def compute_z_sdxl_text_encoders(pipe, request: Dict, hparams, layers, device=None, resolution: int = 512, rng_device=None):
    """(v*, v*_2) of one concept for SDXL's two text encoders: the hidden states of ``layer_module_tmp.format(layer)`` /
    ``.format(layer_2)`` at the last subject token of the first prompt plus vectors found by ONE Adam over both (reference:
    emcid/compute_z.py:651-1037; same arguments, same return).  The UNet sees the concatenated PENULTIMATE hidden states of the
    two encoders as context and the second encoder's ``text_embeds`` + the size ids as added conditions; the loss is the MSE
    between its predictions under the edited source and under the destination embeddings (or the sampled noise), both weight
    decays and, with ``cal_text_repr_loss``, the alignment of both pooled outputs.

    Reference behaviours kept because they decide the result: the destination forward of the SECOND encoder is fed the FIRST
    tokenizer's ids (:842-843); the schedule is ``pipe.scheduler`` (:741).  Results-identical restructuring as in
    compute_z_text_encoder: forward hooks that return an edited copy instead of in-place writes; the clean destination
    forwards and the size ids (loop invariants the reference recomputes every step) run once; the penultimate hidden state is
    taken at the input of each encoder's last layer instead of through ``output_hidden_states``; no loss log file."""
    from PIL import Image
    hp = hparams
    if getattr(hp, "use_ewc", False):
        raise ValueError("ewc not implemented for sdxl")          # reference :951
    te1, te2 = pipe.text_encoder, pipe.text_encoder_2
    dev = next(te1.parameters()).device
    rdev = torch.device(rng_device) if rng_device is not None else dev
    host_draw = rdev.type == "cpu" and dev.type != "cpu"
    layer, layer_2 = layers
    objective = hp.objective
    if objective not in ("ablate-source", "ablate-dest"):
        raise ValueError(f"Objective {objective} can not be used for compute_z.")
    sched = pipe.scheduler
    source_prompts = [p.format(request["source"]) for p in request["prompts"]]
    dest_prompts = [p.format(request["dest"]) for p in request["prompts"]]
    spp = hp.samples_per_prompt
    if "training_img_paths" in request:
        images = [Image.open(path) for path in request["training_img_paths"]]
    elif "images" in request:
        images = request["images"]
    else:
        gen = torch.Generator(dev).manual_seed(int(request["seed_train"])) if request.get("seed_train") is not None else None
        images = []
        with torch.no_grad():
            if objective == "ablate-source":
                for _ in range(spp):
                    images.extend(pipe(source_prompts, guidance_scale=7.5, generator=gen).images)
            else:       # one prompt at a time (:776-779)
                for _ in range(spp):
                    for prompt in source_prompts:
                        images.append(pipe(prompt, guidance_scale=7.5, generator=gen).images[0])
    pixels = preprocess_img(images, resolution)
    bsz = len(source_prompts)
    pixels = pixels.reshape(spp, bsz, *pixels.shape[1:]).transpose(0, 1)
    if len(pixels) % bsz:
        raise AssertionError(f"len(img_batch) {len(pixels)} should be n times of batch size {bsz}")
    tok1, tok2 = pipe.tokenizer, pipe.tokenizer_2
    src_inp, dst_inp = tokenize_prompts(source_prompts, tok1, dev), tokenize_prompts(dest_prompts, tok1, dev)
    src_inp_2, dst_inp_2 = tokenize_prompts(source_prompts, tok2, dev), tokenize_prompts(dest_prompts, tok2, dev)
    f1, f2 = finder_for(tok1), finder_for(tok2)
    src_lookup = [f1(ids, request["source"])[-1] - 1 for ids in src_inp["input_ids"].tolist()]
    src_lookup_2 = [f2(ids, request["source"])[-1] - 1 for ids in src_inp_2["input_ids"].tolist()]
    for ids in dst_inp["input_ids"].tolist():        # the reference looks the destination up too (:818-829): a ValueError if absent
        f1(ids, request["dest"])
    for ids in dst_inp_2["input_ids"].tolist():
        f2(ids, request["dest"])
    if not (len(src_inp["input_ids"]) == len(dst_inp["input_ids"]) == len(pixels)):
        raise AssertionError("The number of prompts and images should be the same.")
    ar = torch.arange(bsz, device=dev)
    idx1, idx2 = torch.tensor(src_lookup, device=dev), torch.tensor(src_lookup_2, device=dev)
    frozen = [prm for m in (te1, te2, pipe.vae, pipe.unet) for prm in m.parameters() if prm.requires_grad]
    for prm in frozen:
        prm.requires_grad_(False)
    delta = torch.zeros((te1.config.hidden_size,), requires_grad=True, device=dev)
    deltas_2 = torch.zeros((te2.config.hidden_size,), requires_grad=True, device=dev)
    opt = torch.optim.Adam([delta, deltas_2], lr=hp.v_lr)
    state = {"edit": False, "init": [None, None], "penult": [None, None]}

    def edit_hook(k, idx, first, dvec):
        def hook(mod, args, out):
            if not state["edit"]:
                return out
            h = out[0] if isinstance(out, tuple) else out
            if state["init"][k] is None:
                state["init"][k] = h[0, first].detach().clone()
            h = h.clone()
            for i in range(h.shape[0]):          # prompt by prompt: the reference's order (:706-728), see compute_z_text_encoder
                if hp.replace_repr:
                    h[i, idx[i], :] = dvec
                else:
                    h[i, idx[i], :] += dvec
            return (h,) + tuple(out[1:]) if isinstance(out, tuple) else h
        return hook

    def penult_hook(k):           # the input of the LAST layer = hidden_states[-2] (after an edit at any earlier layer)
        def hook(mod, args, kwargs):
            state["penult"][k] = args[0] if args else kwargs["hidden_states"]
        return hook

    def last_layer(te):
        return get_module(te, hp.layer_module_tmp.format(te.config.num_hidden_layers - 1))

    handles = [get_module(te1, hp.layer_module_tmp.format(layer)).register_forward_hook(edit_hook(0, [int(i) for i in src_lookup], src_lookup[0], delta)),
               get_module(te2, hp.layer_module_tmp.format(layer_2)).register_forward_hook(edit_hook(1, [int(i) for i in src_lookup_2], src_lookup_2[0], deltas_2)),
               last_layer(te1).register_forward_pre_hook(penult_hook(0), with_kwargs=True),
               last_layer(te2).register_forward_pre_hook(penult_hook(1), with_kwargs=True)]

    def run(te, k, inp, pooled):
        out = te(**inp)
        return state["penult"][k], getattr(out, pooled)

    try:
        with torch.no_grad():       # loop invariants (the reference recomputes them every step from the same frozen encoders)
            dest_txt, dest_pool = run(te1, 0, dst_inp, "pooler_output")
            dest_txt_2, dest_pool_2 = run(te2, 1, dst_inp, "text_embeds")             # the FIRST tokenizer's ids (:842)
            dest_embeds = torch.cat([dest_txt, dest_txt_2], dim=-1)
            height = width = pipe.default_sample_size * pipe.vae_scale_factor
            try:
                add_time_ids = pipe._get_add_time_ids((height, width), (0, 0), (height, width), dtype=dest_embeds.dtype,
                                                      text_encoder_projection_dim=te2.config.projection_dim)
            except (AttributeError, TypeError):      # a pipeline without that helper: diffusers' own layout
                add_time_ids = torch.tensor([[height, width, 0, 0, height, width]], dtype=dest_embeds.dtype)
            add_time_ids = add_time_ids.repeat(bsz, 1).to(dev)
            dest_cond = {"text_embeds": dest_pool_2, "time_ids": add_time_ids}
        posteriors = {}
        for it in range(hp.v_num_grad_steps):
            opt.zero_grad()
            sample_indices = torch.randint(0, spp, (bsz,))
            key = tuple(sample_indices.tolist())
            if key not in posteriors:
                with torch.no_grad():
                    posteriors[key] = pipe.vae.encode(pixels[torch.arange(bsz), sample_indices].to(dev)).latent_dist
            with torch.no_grad():
                latents = posteriors[key].sample(torch.default_generator) if host_draw else posteriors[key].sample()
                latents = latents * pipe.vae.config.scaling_factor
            if host_draw:
                noise = torch.randn(latents.shape, dtype=latents.dtype).to(dev)
                timesteps = torch.randint(0, sched.config.num_train_timesteps, (bsz,)).long().to(dev)
            else:
                noise = torch.randn_like(latents, device=dev)
                timesteps = torch.randint(0, sched.config.num_train_timesteps, (bsz,), device=dev).long()
            noisy = sched.add_noise(latents, noise, timesteps)
            state["edit"] = True
            try:
                edit_txt, edit_pool = run(te1, 0, src_inp, "pooler_output")
                edit_txt_2, edit_pool_2 = run(te2, 1, src_inp_2, "text_embeds")
            finally:
                state["edit"] = False
            edit_embeds = torch.cat([edit_txt, edit_txt_2], dim=-1)
            edit_cond = {"text_embeds": edit_pool_2, "time_ids": add_time_ids}
            if not hp.no_noise_loss:
                edit_pred = pipe.unet(noisy, timesteps, encoder_hidden_states=edit_embeds, added_cond_kwargs=edit_cond).sample
                with torch.no_grad():
                    pred_dest = pipe.unet(noisy, timesteps, encoder_hidden_states=dest_embeds, added_cond_kwargs=dest_cond).sample
            init1, init2 = state["init"]
            reg = hp.v_weight_decay * (torch.norm(delta) / torch.norm(init1) ** 2)
            reg_2 = hp.v_weight_decay * (torch.norm(deltas_2) / torch.norm(init2) ** 2)
            if getattr(hp, "use_sampled_noise", False) or request.get("use_real_noise", False):
                loss = F.mse_loss(noise, edit_pred, reduction="mean") + reg + reg_2
            elif hp.no_noise_loss:
                loss = reg + reg_2
            else:
                loss = F.mse_loss(edit_pred, pred_dest, reduction="mean") + reg + reg_2
            if hp.cal_text_repr_loss and request.get("txt_align", True):
                scale = hp.text_repr_loss_scale_factor
                loss = loss + scale * F.mse_loss(edit_pool, dest_pool, reduction="mean") \
                    + scale * F.mse_loss(edit_pool_2, dest_pool_2, reduction="mean")
            loss.backward()
            opt.step()
            for dvec, init in ((delta, init1), (deltas_2, init2)):
                max_norm = hp.clamp_norm_factor * init.norm()
                if dvec.norm() > max_norm:
                    with torch.no_grad():
                        dvec[...] = dvec * max_norm / dvec.norm()
    finally:
        for hd in handles:
            hd.remove()
        for prm in frozen:
            prm.requires_grad_(True)
    return (state["init"][0] + delta).detach(), (state["init"][1] + deltas_2).detach()


def stage1_for_sdxl(pipe, hparams, layers=None, **kw):
    """``stage1(request, suffix) -> v*`` for an SDXL pipeline: the pair optimisation runs ONCE per request — the reference
    computes and caches both vectors together (emcid_main.py:1157-1230) — and is kept until both encoders' loaders (suffix ""
    for text_encoder, "_2" for text_encoder_2) have asked for their half."""
    layers = layers if layers is not None else (hparams.layers[-1], hparams.layers_2[-1])
    pending: Dict[int, list] = {}

    def stage1(request, suffix=""):
        key = id(request)
        if key not in pending:
            v1, v2 = compute_z_sdxl_text_encoders(pipe, request, hparams, layers, **kw)
            pending[key] = [request, {"": v1, "_2": v2}]
        halves = pending[key][1]
        v = halves.pop(suffix)
        if not halves:
            del pending[key]
        return v

    return stage1


def compute_z_text_encoder_batched(pipe, requests: Sequence[Dict], hparams, layer: int, device=None, noise_scheduler=None,
                                   resolution: int = 512, rng_device=None, batch_size: int = 8) -> List[torch.Tensor]:
    """``[compute_z_text_encoder(pipe, r, ...) for r in requests]`` with ``batch_size`` concepts per Adam step: ONE hooked
    text-encoder forward and ONE UNet forward / backward over the stacked prompts of the concepts (batched
    across concepts Stage 1 is the wall-clock of a real mass edit; the reference runs it per request from the loop at
    emcid/emcid_main.py:871-969).  Every concept keeps its own ``delta`` (one row of a (B, hidden) Adam parameter: Adam is
    element-wise), its own loss terms, norm clamp and — what makes the result that of B sequential calls — its own random
    draws: before the optimisation the draws of concept 1 (image flips; per step sample indices, VAE posterior noise, latent
    noise, timesteps), then those of concept 2, ... are taken in exactly the order sequential calls would take them and kept.
    The UNet and the encoder treat batch rows independently, so d(sum of losses)/d(delta_c) is concept c's own gradient;
    what differs from sequential calls is fp32 rounding inside differently shaped GEMMs.  Concepts are stacked only with
    concepts whose tokenized prompts have the same padded lengths (the UNet sees every position of the sequence)."""
    from PIL import Image
    hp = hparams
    te = pipe.text_encoder
    dev = next(te.parameters()).device
    fim = load_fim(dev) if getattr(hp, "use_ewc", False) else None
    rdev = torch.device(rng_device) if rng_device is not None else dev
    host_draw = rdev.type == "cpu" and dev.type != "cpu"
    tok = pipe.tokenizer
    sched = noise_scheduler if noise_scheduler is not None else default_noise_scheduler()
    objective = hp.objective
    if objective not in ("ablate-source", "ablate-dest", "esd"):
        raise ValueError(f"Objective {objective} can not be used for compute_z.")
    spp, steps = hp.samples_per_prompt, hp.v_num_grad_steps
    finder = finder_for(tok)
    results: List[Optional[torch.Tensor]] = [None] * len(requests)
    frozen = [prm for m in (te, pipe.vae, pipe.unet) for prm in m.parameters() if prm.requires_grad]
    for prm in frozen:
        prm.requires_grad_(False)
    mod = get_module(te, hp.layer_module_tmp.format(layer))
    try:
        for lo in range(0, len(requests), max(1, int(batch_size))):
            chunk = list(range(lo, min(len(requests), lo + max(1, int(batch_size)))))
            # ---- phase A: per concept, in request order: inputs, loop invariants and ALL random draws ---------------------
            ctxs = []
            for ri in chunk:
                request = requests[ri]
                c = {"ri": ri, "request": request}
                source_prompts = [p.format(request["source"]) for p in request["prompts"]]
                dest_prompts = ["" for _ in request["prompts"]] if objective == "esd" else [p.format(request["dest"]) for p in request["prompts"]]
                if "training_img_paths" in request and objective != "esd":
                    images = [Image.open(path) for path in request["training_img_paths"]]
                elif "images" in request and objective != "esd":
                    images = request["images"]
                else:
                    gen = torch.Generator(dev).manual_seed(int(request["seed_train"])) if request.get("seed_train") is not None else None
                    images = []
                    for _ in range(spp):
                        images.extend(pipe(source_prompts, guidance_scale=7.5, generator=gen).images)
                pixels = preprocess_img(images, resolution)
                bsz = len(source_prompts)
                pixels = pixels.reshape(spp, bsz, *pixels.shape[1:]).transpose(0, 1)
                if len(pixels) % bsz:
                    raise AssertionError(f"len(img_batch) {len(pixels)} should be n times of batch size {bsz}")
                src_inp, dst_inp = tokenize_prompts(source_prompts, tok, dev), tokenize_prompts(dest_prompts, tok, dev)
                src_lookup = [finder(ids, request["source"])[-1] - 1 for ids in src_inp["input_ids"].tolist()]
                dst_lookup = [finder(ids, request["dest"])[-1] - 1 for ids in dst_inp["input_ids"].tolist()]
                if not (len(src_inp["input_ids"]) == len(dst_inp["input_ids"]) == len(pixels)):
                    raise AssertionError("The number of prompts and images should be the same.")
                c.update(bsz=bsz, src_inp=src_inp, src_lookup=src_lookup, dst_lookup=dst_lookup)
                with torch.no_grad():
                    c["dest_repr"], c["dest_pool"] = te(**dst_inp)[0:2]
                    c["source_repr"] = te(**src_inp)[0] if (objective == "esd" or hp.cal_text_repr_loss) else None
                    if hp.contrastive_text_loss:
                        c["neg_pool"] = te(**tokenize_prompts(request["negative_prompts"], tok, dev))[1]
                        c["single_pool"] = te(**tokenize_prompts([request["dest"]], tok, dev))[1]
                    if hp.align_obj_eos_pad:
                        full = lambda ps: {k: v.to(dev) for k, v in tok(ps, max_length=tok.model_max_length, return_tensors="pt",
                                                                        padding="max_length", truncation=True).items()}
                        src_full, dst_full = full(source_prompts), full(dest_prompts)
                        src_eos = [int(m.sum()) - 1 for m in src_full["attention_mask"]]
                        dst_eos = [int(m.sum()) - 1 for m in dst_full["attention_mask"]]
                        far = max(src_eos + dst_eos)
                        c["src_full"] = src_full
                        c["src_slices"] = [list(range(e, tok.model_max_length - max(0, far - e))) for e in src_eos]
                        dst_slices = [list(range(e, tok.model_max_length - max(0, far - e))) for e in dst_eos]
                        dest_full = te(**dst_full)[0]
                        c["d_pad"] = torch.stack([dest_full[i, sl, :] for i, sl in enumerate(dst_slices)], dim=0)
                # the draws of every step, in the order compute_z_text_encoder takes them
                posteriors, draws = {}, []
                for _ in range(steps):
                    sample_indices = torch.randint(0, spp, (bsz,))
                    key = tuple(sample_indices.tolist())
                    if key not in posteriors:
                        with torch.no_grad():
                            posteriors[key] = pipe.vae.encode(pixels[torch.arange(bsz), sample_indices].to(dev)).latent_dist
                    with torch.no_grad():
                        latents = posteriors[key].sample(torch.default_generator) if host_draw else posteriors[key].sample()
                        latents = latents * pipe.vae.config.scaling_factor
                    if host_draw:
                        noise = torch.randn(latents.shape, dtype=latents.dtype).to(dev)
                        timesteps = torch.randint(0, sched.config.num_train_timesteps, (bsz,)).long().to(dev)
                    else:
                        noise = torch.randn_like(latents, device=dev)
                        timesteps = torch.randint(0, sched.config.num_train_timesteps, (bsz,), device=dev).long()
                    draws.append((sched.add_noise(latents, noise, timesteps), noise, timesteps))
                c["draws"] = draws
                ctxs.append(c)
            # ---- phase B: concepts with equal padded prompt lengths share the forward / backward passes --------------------
            groups: Dict[tuple, List[dict]] = {}
            for c in ctxs:
                groups.setdefault((c["src_inp"]["input_ids"].shape[1], c["dest_repr"].shape[1]), []).append(c)
            for members in groups.values():
                _stage1_optimise_group(pipe, te, mod, hp, members, steps, dev, fim)
                for c in members:
                    results[c["ri"]] = c["v_star"]
    finally:
        for prm in frozen:
            prm.requires_grad_(True)
    return results


def _stage1_optimise_group(pipe, te, mod, hp, members, steps, dev, fim=None):
    """The Adam loop of compute_z_text_encoder for several concepts at once (rows of concept c: [off[c], off[c + 1]))."""
    objective = hp.objective
    B = len(members)
    sizes = [c["bsz"] for c in members]
    off = np.cumsum([0] + sizes)
    rows = int(off[-1])
    owner = torch.tensor(np.repeat(np.arange(B), sizes), device=dev)                   # concept of every stacked row
    ar = torch.arange(rows, device=dev)
    src_idx_host = [int(i) for c in members for i in c["src_lookup"]]
    owner_host = np.repeat(np.arange(B), sizes).tolist()
    src_idx = torch.tensor(src_idx_host, device=dev)
    dst_idx = torch.tensor([i for c in members for i in c["dst_lookup"]], device=dev)
    cat = lambda key: {k: torch.cat([c[key][k] for c in members], dim=0) for k in members[0][key]}
    src_inp = cat("src_inp")
    dest_repr = torch.cat([c["dest_repr"] for c in members], dim=0)
    dest_pool = torch.cat([c["dest_pool"] for c in members], dim=0)
    source_repr = torch.cat([c["source_repr"] for c in members], dim=0) if members[0]["source_repr"] is not None else None
    src_full = cat("src_full") if hp.align_obj_eos_pad else None
    delta = torch.zeros((B, te.config.hidden_size), requires_grad=True, device=dev)
    opt = torch.optim.Adam([delta], lr=hp.v_lr)
    state = {"edit": False, "source_init": None}

    def hook(module, args, out):
        if not state["edit"]:
            return out
        h = out[0] if isinstance(out, tuple) else out
        if state["source_init"] is None:          # per concept: the clean state at the lookup token of its FIRST prompt
            first = torch.tensor(off[:-1], device=dev)
            state["source_init"] = h[first, src_idx[first]].detach().clone()
        h = h.clone()
        for r in range(rows):                # row by row in every concept's own prompt order: delta_c's gradient is summed over its
            c = owner_host[r]                # prompts in the order a sequential call sums it (see compute_z_text_encoder)
            if hp.replace_repr:
                h[r, src_idx_host[r], :] = delta[c]
            else:
                h[r, src_idx_host[r], :] += delta[c]
        return (h,) + tuple(out[1:]) if isinstance(out, tuple) else h

    def edited(inp):
        state["edit"] = True
        try:
            return te(**inp)[0:2]
        finally:
            state["edit"] = False

    def per_concept_mse(a, b):
        """[F.mse_loss(a[rows of c], b[rows of c]) for c]: every row has the same number of elements."""
        per_row = ((a - b) ** 2).reshape(rows, -1).mean(dim=1)
        return torch.zeros(B, device=dev, dtype=per_row.dtype).index_add(0, owner, per_row) / torch.tensor(sizes, device=dev, dtype=per_row.dtype)

    handle = mod.register_forward_hook(hook)
    try:
        for it in range(steps):
            opt.zero_grad()
            noisy = torch.cat([c["draws"][it][0] for c in members], dim=0)
            noise = torch.cat([c["draws"][it][1] for c in members], dim=0)
            timesteps = torch.cat([c["draws"][it][2] for c in members], dim=0)
            edit_repr, edit_pool = edited(src_inp)
            source_init = state["source_init"]                                       # (B, hidden)
            if not hp.no_noise_loss:
                edit_pred = pipe.unet(noisy, timesteps, edit_repr).sample
                with torch.no_grad():
                    pred_dest = pipe.unet(noisy, timesteps, dest_repr).sample
            if fim is not None and "ablate" in objective:        # EWC instead of the weight decay, per concept (:547-549)
                reg = torch.sum(float(hp.ewc_lambda) * fim * delta ** 2, dim=1) / (2 * torch.norm(source_init, dim=1) ** 2)
            else:
                reg = hp.v_weight_decay * (torch.norm(delta, dim=1) / torch.norm(source_init, dim=1) ** 2)       # (B,)
            if "ablate" in objective:
                if getattr(hp, "use_sampled_noise", False):
                    loss = per_concept_mse(noise, edit_pred) + reg
                elif hp.no_noise_loss:
                    loss = reg
                else:
                    real = [bool(c["request"].get("use_real_noise", False)) for c in members]
                    loss = per_concept_mse(edit_pred, pred_dest) + reg
                    if any(real):
                        sel = torch.tensor(real, device=dev)
                        loss = torch.where(sel, per_concept_mse(noise, edit_pred) + reg, loss)
            else:
                with torch.no_grad():
                    pred_source = pipe.unet(noisy, timesteps, source_repr).sample
                loss = per_concept_mse(edit_pred, pred_dest - hp.esd_mu * (pred_source - pred_dest)) + reg
            if hp.cal_text_repr_loss:
                scale = hp.text_repr_loss_scale_factor
                align = torch.tensor([bool(c["request"].get("txt_align", True)) for c in members], device=dev)
                if hp.contrastive_text_loss:
                    terms = []
                    for ci, c in enumerate(members):
                        emb = torch.cat([c["single_pool"], c["neg_pool"]], dim=0)
                        scores = torch.squeeze(-torch.cdist(edit_pool[off[ci]:off[ci + 1]].unsqueeze(0), emb.unsqueeze(0)))
                        terms.append(-torch.log_softmax(scores, dim=1)[:, 0].mean(dim=0))
                    term = torch.stack(terms)
                elif hp.align_object_token:
                    term = per_concept_mse(edit_repr[ar, src_idx, :], dest_repr[ar, dst_idx, :])
                elif hp.align_obj_eos_pad:
                    e_full = edited(src_full)[0]
                    terms = []
                    for ci, c in enumerate(members):
                        r0 = int(off[ci])
                        e_pad = torch.stack([e_full[r0 + i, sl, :] for i, sl in enumerate(c["src_slices"])], dim=0)
                        sl_rows = slice(r0, int(off[ci + 1]))
                        terms.append(F.mse_loss(torch.cat([edit_repr[ar[sl_rows], src_idx[sl_rows], :].unsqueeze(1), e_pad], dim=1),
                                                torch.cat([dest_repr[ar[sl_rows], dst_idx[sl_rows], :].unsqueeze(1), c["d_pad"]], dim=1),
                                                reduction="mean"))
                    term = torch.stack(terms)
                else:
                    term = per_concept_mse(edit_pool, dest_pool)
                loss = loss + torch.where(align, scale * term, torch.zeros_like(term))
            loss.sum().backward()
            opt.step()
            with torch.no_grad():
                max_norm = hp.clamp_norm_factor * source_init.norm(dim=1)
                nrm = delta.norm(dim=1)
                factor = torch.where(nrm > max_norm, max_norm / nrm, torch.ones_like(nrm))
                delta.mul_(factor.unsqueeze(1))
    finally:
        handle.remove()
    out = (state["source_init"] + delta).detach()
    for ci, c in enumerate(members):
        c["v_star"] = out[ci].clone()
        c.pop("draws", None)


def stage1_for(pipe, hparams, layer: int, batch_size: Optional[int] = None, clip_towers=None, **kw):
    """The ``stage1=`` callable emcid_main's v* cache expects (``stage1(request, suffix) -> v*``) for a pipeline that carries
    a UNet and a VAE: Stage 1 on a cache miss, like the reference (emcid_main.py:905-969).  Its ``batch(requests, suffix)``
    attribute serves all misses of a request list at once (compute_z_text_encoder_batched; EMCID_STAGE1_BATCH concepts per
    Adam step, default 8; 1 = one concept at a time)."""
    new_z = bool(getattr(hparams, "use_new_compute_z", False))      # (num_edit_tokens, hidden) per concept (emcid_main.py:927-936)
    sld = bool(getattr(hparams, "sld_supervision", False))          # the global-concept form (emcid_main.py:911-918)
    v1 = getattr(hparams, "txt_img_align_scale_factor", 0) != 0     # the projected-space form with CLIP's towers (emcid_main.py:919-926)

    def stage1(request, suffix=""):
        if suffix:
            raise ValueError("a suffixed v* belongs to SDXL's second encoder: use stage1_for_sdxl (compute_z_sdxl_text_encoders)")
        if sld:              # the reference's order of the three tests (emcid_main.py:911-936)
            return compute_z_text_encoder_global(pipe, request, hparams, layer, **kw)
        if v1:
            return compute_z_text_encoder_v1(pipe, request, hparams, layer, clip_towers=clip_towers, **kw)
        if new_z:
            return compute_z_text_encoder_v2(pipe, request, hparams, layer, **kw)
        return compute_z_text_encoder(pipe, request, hparams, layer, **kw)

    def batch(requests, suffix=""):
        if suffix:
            raise ValueError("a suffixed v* belongs to SDXL's second encoder: use stage1_for_sdxl (compute_z_sdxl_text_encoders)")
        if sld or v1:
            return [stage1(r, suffix) for r in requests]
        if new_z:
            return [compute_z_text_encoder_v2(pipe, r, hparams, layer, **kw) for r in requests]
        import os
        bs = batch_size if batch_size is not None else int(os.environ.get("EMCID_STAGE1_BATCH", "8"))
        if bs <= 1:
            return [compute_z_text_encoder(pipe, r, hparams, layer, **kw) for r in requests]
        return compute_z_text_encoder_batched(pipe, requests, hparams, layer, batch_size=bs, **kw)

    stage1.batch = batch
    return stage1
