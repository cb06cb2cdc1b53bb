"""K/Z assembly: fc2 input ("key") and fc2 output at each concept's last subject token.

Host-side counterpart of the slice of the reference's emcid/compute_z.py that Stage 2 calls:
``tokenize_prompts`` (:56-74) and ``get_module_input_output_at_words`` (:2252-2384).  Stage 1
(``compute_z_text_encoder*``, Adam through the UNet) is out of scope: its OUTPUT, the cached ``v_star``
npz, is an input here.

MI355X-first differences, results identical:
* the prompt batch (ids, mask, lookup index per prompt, request segment offsets) is built once on the host
  and lives in HBM (``PromptBatch``); the reference re-tokenizes and re-searches on every call;
* the N*P Python indexings and N Python means are one gather+mean kernel (csrc/gram_f32.hip,
  ``emcid_gather_mean_f32``), summed in prompt order with a true division so it is bit-compatible with
  torch-CPU's ``.mean(0)``;
* the forward stops at the hooked module (the reference runs the remaining layers and discards them).
"""
import weakref
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip
from .causal_trace import TokenRangeFinder
from .clip_attention import hip_attention
from .nethook import StopForward, get_module


def tokenize_prompts(prompts, tokenizer, device, padding_length=None):
    if padding_length is None:
        enc = tokenizer(prompts, return_tensors="pt", padding=True, truncation=True)
    else:
        enc = tokenizer(prompts, return_tensors="pt", padding="max_length", truncation=True, max_length=padding_length)
    return {k: v.to(device) for k, v in enc.items()}


def expand_request_prompts(requests: Sequence[Dict]) -> Tuple[List[str], List[str], List[int]]:
    """Flattened prompt strings, the subject of each, and prompts-per-request.
    ``source_prompts`` (pre-formatted) wins over ``prompts`` when the FIRST request carries it, and the
    per-request count is taken from ``prompts`` when the first request has that key — both as the reference
    does (compute_z.py:2270-2283, :2318-2320)."""
    first = requests[0]
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

    @property
    def n_prompts(self):
        return int(self.lookup.numel())


def tokenize_lists(tokenizer, prompts: Sequence[str]) -> Dict[str, np.ndarray]:
    """``tokenizer(prompts, padding=True, truncation=True)`` as (B, S) int64 arrays (reference: compute_z.py:65).

    For a tokenizers-backed HF tokenizer the public call spends most of its time assembling a ``BatchEncoding`` and
    tracking character offsets nobody reads.  Here a ONE-prompt public call configures the backend's truncation and
    padding exactly as transformers does for these arguments (``set_truncation_and_padding``), then the backend encodes
    the batch without offsets (``encode_batch_fast``; same ids and masks by construction) — and the one-prompt public
    result is checked against the corresponding row.  Anything unexpected falls back to the public call."""
    bt = getattr(tokenizer, "_tokenizer", None)
    if bt is not None and hasattr(bt, "encode_batch_fast") and len(prompts) > 8:
        try:
            longest = max(range(len(prompts)), key=lambda i: len(prompts[i]))
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


def gather_request_means(act: torch.Tensor, batch: PromptBatch) -> torch.Tensor:
    """(B, S, c) activations -> (N, c): row at each prompt's lookup token, averaged per request."""
    if act.dtype != torch.float32:
        raise hip.EmcidHipError(f"K/Z assembly is fp32 (reference loads the encoder in fp32); got {act.dtype}")
    if act.stride(-1) != 1:
        act = act.contiguous()
    return hip.gather_mean(act, batch.lookup, batch.seg)


def get_module_input_output_at_words(text_encoder, tok, requests: List[Dict], module_name: str,
                                     num_fact_token: int = 1, batch: Optional[PromptBatch] = None
                                     ) -> Tuple[torch.Tensor, torch.Tensor]:
    """(input_ret (N, d), output_ret (N, h)) of ``module_name`` at the last subject token, mean over each
    request's prompts (reference: compute_z.py:2252-2325, the ``num_fact_token == 1`` branch)."""
    if num_fact_token != 1:
        raise NotImplementedError("num_fact_token > 1 (compute_z.py:2329-2382) is unused by every shipped hparams "
                                  "file and is not built")
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
