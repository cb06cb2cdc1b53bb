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

    @property
    def n_prompts(self):
        return int(self.lookup.numel())


def build_prompt_batch(tokenizer, requests: Sequence[Dict], device, finder: Optional[TokenRangeFinder] = None,
                       truncate: bool = True) -> PromptBatch:
    prompts, subjects, counts = expand_request_prompts(requests)
    # plain lists from the tokenizer, tensors built here: `return_tensors="pt"` walks every id in Python (a third of
    # the host time of this function at 3 000 prompts) and the lists are needed for the subject search anyway
    enc_lists = tokenizer(prompts, padding=True, truncation=True)
    enc = {k: torch.tensor(v, dtype=torch.int64) for k, v in enc_lists.items()}
    finder = finder or TokenRangeFinder(tokenizer)
    ids_host = enc_lists["input_ids"]
    lookup = [r[-1] - 1 for r in finder.batch(ids_host, subjects)]
    if len(ids_host) != len(lookup):
        raise ValueError("The number of prompts and lookup indices should be the same.")
    S = enc["input_ids"].shape[1]
    for i, j in enumerate(lookup):
        if not 0 <= j < S:
            raise ValueError(f"lookup index {j} outside the padded prompt (S={S}) for prompt {prompts[i]!r}")
    seg = np.cumsum([0] + counts)
    if seg[-1] != len(prompts):
        raise ValueError(f"request prompt counts ({seg[-1]}) do not cover the {len(prompts)} prompts")
    if truncate:
        # CLIP text attention is causal: nothing at or before a lookup token depends on later positions, so
        # the columns after the last lookup index (EOS, padding) are never needed by the K/Z gather.
        keep = max(lookup) + 1
        enc = {k: v[:, :keep] for k, v in enc.items()}
    return PromptBatch(
        inputs={k: v.to(device) for k, v in enc.items()},
        lookup=torch.tensor(lookup, dtype=torch.int64, device=device),
        seg=torch.tensor(seg, dtype=torch.int64, device=device),
        n_requests=len(requests), lookup_host=lookup)


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
