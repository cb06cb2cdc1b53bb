"""Keys of one layer for a request list (reference: emcid/compute_ks.py:21-41 ``compute_ks_text_encoder``) and the
keys / current values of the UNet's cross-attention projections (:52-141)."""
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import clip_forward, hip
from .clip_attention import hip_attention
from .compute_z import PromptBatch, build_prompt_batch, get_module_input_output_at_words
from .nethook import get_module


def compute_ks_text_encoder(model, tok, requests: List[Dict], hparams, layer: int):
    """(num_requests, d): mean fc2-input at the last subject token of each request's prompts."""
    layername = hparams.rewrite_module_tmp.format(layer)
    return get_module_input_output_at_words(model, tok, requests, layername,
                                            num_fact_token=hparams.num_edit_tokens)[0]


def text_embedding_at_lookup(pipe, batch: PromptBatch, layer_module_tmp: Optional[str] = None) -> torch.Tensor:
    """(B, hidden): the text encoder's last hidden state at each prompt's last subject token."""
    te = pipe.text_encoder
    if layer_module_tmp is not None and batch.lookup.is_cuda:
        try:
            graph = clip_forward.discover(te, layer_module_tmp)
            trie = clip_forward.build_trie(batch.ids_host, batch.lookup_host, batch.lookup.device)
            with torch.no_grad():
                out = clip_forward.last_hidden_at_lookup(graph, trie)
            clip_forward.LAST_PATHS["forward_trie"] += 1
            return out
        except (clip_forward.UnsupportedEncoder, IndexError, LookupError) as e:
            clip_forward.note_fallback("text_embedding_at_lookup", e)       # counted and logged, never silent
    clip_forward.LAST_PATHS["forward_hf"] += 1
    with torch.no_grad(), hip_attention(te):
        rep = te(**batch.inputs)[0]
    return rep[torch.arange(rep.shape[0], device=rep.device), batch.lookup]


def get_layers_input_output_at_words_cross_attn(pipe, requests: List[Dict], module_names: Sequence[str],
                                                batch: Optional[PromptBatch] = None,
                                                layer_module_tmp: Optional[str] = None
                                                ) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    """({name: (N, hidden)}, {name: (N, out)}) — input and output of every ``attn2.to_k`` / ``to_v`` at the last subject
    token, mean over each request's prompts (reference: compute_ks.py:52-141).

    The reference runs the whole UNet on dummy latents once per request only to hook these projections; their input
    IS the text embedding, so here the encoder runs once (prefix trie when it is a HF CLIP text model) and each
    projection is one GEMM on the B lookup rows.  Every request must have the same number of prompts (:66, :78)."""
    device = next(pipe.text_encoder.parameters()).device
    counts = {len(r["source_prompts"] if "source_prompts" in requests[0] else r["prompts"]) for r in requests}
    assert len(counts) == 1, "All the requests should have the same number of prompts."
    if batch is None:
        batch = build_prompt_batch(pipe.tokenizer, requests, device)
    rows = text_embedding_at_lookup(pipe, batch, layer_module_tmp).contiguous()            # (B, hidden)
    B = rows.shape[0]
    zero = torch.zeros(B, dtype=torch.int64, device=device)

    def request_means(x):                                                                   # (B, c) -> (N, c)
        return hip.gather_mean(x.unsqueeze(1), zero, batch.seg) if x.is_cuda else torch.stack(
            [x[a:b].mean(0) for a, b in zip(batch.seg[:-1].tolist(), batch.seg[1:].tolist())], 0)

    keys = request_means(rows)
    ins, outs = {}, {}
    with torch.no_grad():
        for name in module_names:
            mod = get_module(pipe.unet, name)
            ins[name] = keys
            # one GEMM per projection on the lookup rows, on the library's kernel (clip_forward.linear counts the path it took)
            outs[name] = request_means(clip_forward.linear(rows, mod.weight, mod.bias).contiguous())
    return ins, outs
