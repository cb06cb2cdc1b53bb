"""UCE closed-form edit — the baseline the reference ships beside EMCID (emcid/uce_train.py), on the same HIP kernels.

    W_new = (lam W + e * sum_edits V^T K + p * sum_retain Vr^T Kr) (lam I + e * sum K^T K + p * sum Kr^T Kr)^-1

Same entry points and argument meaning as the reference:

* ``edit_text_encoder_uce`` (uce_train.py:31-213): one fc2 of the text encoder; K = fc2 inputs, V from the fc2 module.
* ``edit_model_uce`` (uce_train.py:216-416): the cross-attention to_v / to_k projections of the UNet; K = final text
  embeddings, V from each projection's own weight.

What differs from the reference is the schedule and the arithmetic after the encoder, not the result:

* every text (old, new, retain) goes through the encoder ONCE, batched; the reference re-runs the encoder on a batch of
  two for every (edit, projection) pair (32 x N forwards for SD's UNet);
* the sums are fp64 MFMA GEMMs over all rows at once (``emcid_dgemm_f64`` / ``emcid_dgemm_ex_f64`` lower-only Gram), not
  per-row outer products summed in fp32; the normal matrix is SPD, so ``mat1 @ inverse(mat2)`` is a Cholesky
  factorization + two triangular solves (``emcid_cholesky_f64`` / ``emcid_cholesky_solve_f64``), fp64, instead of an fp32
  ``torch.inverse``;
* cross-attention: the normal matrix depends on the texts only, so it is factored once for all projections (the
  reference inverts it per projection), and all projections' right-hand sides are solved in one call.

The fp32 inverse costs the reference ~2e-4 (relative, toy fixture); the tests hold this path to the oracle's fp64 mode
tightly and to the reference-minted fixture within that error.  No CPU fallback: ``hip`` raises without the library/GPU.
"""
from __future__ import annotations

import ast
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import clip_forward, hip, nethook
from .layer_stats import get_all_cross_attn_kv_layer_names

FORWARD_CHUNK = 512          # texts per encoder forward (x model_max_length tokens each)
MAX_VALUE_COLS = 8192        # value rows: projections are processed in groups of at most this many output columns (fp64 M x cols buffers)
EDIT_CHUNK = 1024            # per-edit Grams: edits per batch (2 x chunk x d x d fp64 buffers)
LAST_RUN: Dict[str, float] = {}
# module switches (scripts/bench_uce.py flips them): the rows' forward ("packed" = the prefix-trie forward, else the hooked HF
# forward) and the closed form's solver ("auto", or one of closed_form's methods)
UCE_FORWARD = "packed"
UCE_METHOD = "auto"


# ---- host side: texts -> row windows -------------------------------------------------------------------------------

def _format_texts(old_text_, new_text_, retain_text_) -> Tuple[List[str], List[str], List[str]]:
    """uce_train.py:52-66: '' as a new text becomes ' '; no retain list means the empty prompt is retained."""
    old_texts = list(old_text_)
    new_texts = [(" " if t == "" else t) for t in new_text_]
    if len(old_texts) != len(new_texts):         # the reference zips (silently truncating); a mismatch is a caller bug
        n = min(len(old_texts), len(new_texts))
        old_texts, new_texts = old_texts[:n], new_texts[:n]
    ret_texts = [""] if retain_text_ is None else list(retain_text_)
    return old_texts, new_texts, ret_texts


def _tokenize(tokenizer, texts: Sequence[str]):
    return tokenizer(list(texts), padding="max_length", max_length=tokenizer.model_max_length, truncation=True,
                     return_tensors="pt")


def row_windows(attention_mask: np.ndarray, n_edits: int, n_rows: int):
    """Flat (text, position) indices of the context rows (old text) and the rows the values come from (new text):
    from the last subject token to the end of the padded sequence, the longer text's tail cut so that both windows have
    the same length (uce_train.py:109-127).  Texts are laid out [old_0, new_0, old_1, new_1, ...].
    Returns (old_flat, new_flat, seg) as int64 arrays; seg = edit index of every row."""
    lens = attention_mask.sum(axis=1).astype(np.int64)
    old_flat, new_flat, seg = [], [], []
    for i in range(n_edits):
        f_old, f_new = int(lens[2 * i]) - 2, int(lens[2 * i + 1]) - 2
        far = max(f_old, f_new)
        m = n_rows - far
        old_flat.append((2 * i) * n_rows + f_old + np.arange(m))
        new_flat.append((2 * i + 1) * n_rows + f_new + np.arange(m))
        seg.append(np.full(m, i, dtype=np.int64))
    return np.concatenate(old_flat), np.concatenate(new_flat), np.concatenate(seg)


def _encode_rows_packed(pipe, ids: torch.Tensor, flat_sets: Sequence[np.ndarray], tap_module: Optional[str]):
    """The same rows through the explicit CLIP forward over the prefix trie of the padded token rows (the encoder is
    called with input_ids only, so padding positions are ordinary causal tokens): identical texts — the usual case for
    the new / anchor concept — and shared beginnings are computed once.  Raises UnsupportedEncoder for anything that is
    not an HF CLIP text model in fp32; the caller then takes the hooked forward."""
    dev = pipe.device
    te = pipe.text_encoder
    names = [n for n, _ in te.named_modules() if n.endswith("encoder.layers.0")]
    if not names:
        raise clip_forward.UnsupportedEncoder("no encoder.layers")
    graph = clip_forward.discover(te, names[0][:-1] + "{}")
    trie, _, nodes = clip_forward.build_trie_packed(ids.tolist(), dev, return_nodes=True)
    if tap_module is None:
        if graph.final_layer_norm is None:
            raise clip_forward.UnsupportedEncoder("no final_layer_norm")
        hs = clip_forward.run_layers(graph, trie, len(graph.layers) - 1, None, last_rows_only=False)
        feats = graph.final_layer_norm(hs)
    else:
        mods = dict(te.named_modules())
        layer = next(i for i, l in enumerate(graph.layers) if l.fc2 is mods[tap_module])
        box = {}

        def on_fc2(i, x, out):
            if i == layer:
                box["x"] = x
                return None                      # stop: nothing after the tapped fc2 input is needed
            return out
        clip_forward.run_layers(graph, trie, layer, on_fc2, last_rows_only=False, fc2_by_callback={layer})
        feats = box["x"]
    flat_nodes = torch.from_numpy(nodes.reshape(-1)).to(dev)
    return [feats.index_select(0, flat_nodes.index_select(0, torch.from_numpy(np.asarray(f, dtype=np.int64)).to(dev))).double()
            for f in flat_sets]


def _encode_rows(pipe, ids: torch.Tensor, flat_sets: Sequence[np.ndarray], tap_module: Optional[str]):
    """Context / value-source rows: the packed explicit forward when the encoder allows it, else the hooked HF forward."""
    if UCE_FORWARD == "packed":
        try:
            out = _encode_rows_packed(pipe, ids, flat_sets, tap_module)
            LAST_RUN["forward"] = "packed-trie"
            return out
        except clip_forward.UnsupportedEncoder as e:
            clip_forward.note_fallback("uce_train", e)
    LAST_RUN["forward"] = "hooked"
    return _encode_rows_hooked(pipe, ids, flat_sets, tap_module)


def _encode_rows_hooked(pipe, ids: torch.Tensor, flat_sets: Sequence[np.ndarray], tap_module: Optional[str]):
    """Run the encoder over the token rows `ids` in chunks and return, per index set, the fp64 rows at those flat
    (text*S + pos) positions of either the tapped module's INPUT (fc2 variant) or the encoder's output [0] (cross-attention
    variant).  input_ids only, as the reference calls it (uce_train.py:105, :298)."""
    dev = pipe.device
    S = ids.shape[1]
    outs = [[] for _ in flat_sets]
    sets = [torch.from_numpy(np.asarray(f, dtype=np.int64)) for f in flat_sets]
    for c0 in range(0, ids.shape[0], FORWARD_CHUNK):
        c1 = min(ids.shape[0], c0 + FORWARD_CHUNK)
        chunk = ids[c0:c1].to(dev)
        with torch.no_grad():
            if tap_module is None:
                feats = pipe.text_encoder(chunk)[0]
            else:
                with nethook.Trace(pipe.text_encoder, tap_module, retain_input=True, retain_output=False, stop=True) as tr:
                    pipe.text_encoder(chunk)
                feats = tr.input
        flat = feats.reshape(-1, feats.shape[-1])
        for k, f in enumerate(sets):
            sel = f[(f >= c0 * S) & (f < c1 * S)] - c0 * S
            if sel.numel():
                outs[k].append(flat.index_select(0, sel.to(dev)).double())
    width = flat.shape[1]
    return [torch.cat(o, 0) if o else torch.zeros(0, width, dtype=torch.float64, device=dev) for o in outs]


# ---- device side: the closed form ----------------------------------------------------------------------------------

def _pad128(n: int) -> int:
    return (n + 127) // 128 * 128


def _normal_matrix(Ko: torch.Tensor, Kr: torch.Tensor, lamb: float, e: float, p_eff: float):
    """lam I + e Ko^T Ko + p_eff Kr^T Kr, lower triangle, padded to a multiple of 128 with a unit diagonal; factored.
    Returns (L, inverse-block workspace).  The contraction runs over 10^4..10^5 rows into a few output tiles: with
    beta == 1 the GEMM launcher splits K over workgroups and adds the partial tiles with f64 atomics (gemm_f64.h)."""
    d = Ko.shape[1]
    dp = _pad128(d)
    A = torch.zeros(dp, dp, dtype=torch.float64, device=Ko.device)
    A.diagonal().fill_(1.0)
    A.diagonal()[:d] = lamb
    blk = A[:d, :d]                         # a strided view: the GEMM writes with ldc = dp
    if Ko.shape[0]:
        hip.dgemm_ex(1, 1, Ko, Ko, blk, alpha=e, beta=1.0, flags=16)
    if Kr.shape[0]:
        hip.dgemm_ex(1, 1, Kr, Kr, blk, alpha=p_eff, beta=1.0, flags=16)
    L, inv, info = hip.cholesky(A)
    if int(info.item()) != 0:
        raise hip.EmcidHipError(f"UCE normal matrix is not positive definite (info={int(info.item())}): lamb must be > 0")
    return L, inv


def _values(Ko, Kn, seg, n_edits, Wg, bg, col_group, technique):
    """Value rows of a group of projections stacked along columns (Wg (cols, d) f64; bg (cols,) or None; col_group (cols,)
    = projection index of every column).  'tensor' (uce_train.py:155-166): per edit and projection,
    S = N - <O, N>/<O, O> O with O / N the old / new rows through the projection; otherwise S = N."""
    M = Kn.shape[0]
    Nw = torch.empty(M, Wg.shape[0], dtype=torch.float64, device=Kn.device)
    hip.dgemm(0, 0, Kn, Wg, Nw)
    if bg is not None:
        Nw += bg
    if technique != "tensor":
        return Nw
    O = torch.empty_like(Nw)
    hip.dgemm(0, 0, Ko, Wg, O)
    if bg is not None:
        O += bg
    n_proj = int(col_group.max().item()) + 1
    ind = torch.zeros(Wg.shape[0], n_proj, dtype=torch.float64, device=Kn.device)
    ind[torch.arange(Wg.shape[0], device=Kn.device), col_group] = 1.0
    dot = torch.zeros(n_edits, n_proj, dtype=torch.float64, device=Kn.device).index_add_(0, seg, (O * Nw) @ ind)
    sq = torch.zeros(n_edits, n_proj, dtype=torch.float64, device=Kn.device).index_add_(0, seg, (O * O) @ ind)
    alpha = dot / sq                                         # (edits, projections)
    Nw -= alpha[seg][:, col_group] * O
    return Nw


def _solve_group(Ko, Kn, Kr, seg, n_edits, weights, biases, L, inv, lamb, e, p_eff, technique):
    """New weights of a group of projections that share the context rows (and hence the factored normal matrix)."""
    dev = Ko.device
    d = Ko.shape[1]
    dp = L.shape[0]
    Wg = torch.cat([w.detach().double() for w in weights], 0).contiguous()
    bg = None if biases[0] is None else torch.cat([b.detach().double() for b in biases], 0)
    col_group = torch.cat([torch.full((w.shape[0],), k, dtype=torch.int64, device=dev) for k, w in enumerate(weights)])
    rhs = torch.zeros(Wg.shape[0], dp, dtype=torch.float64, device=dev)     # mat1, padded columns stay 0
    blk = rhs[:, :d]
    blk.copy_(Wg).mul_(lamb)
    if Ko.shape[0]:
        S = _values(Ko, Kn, seg, n_edits, Wg, bg, col_group, technique)
        hip.dgemm_ex(1, 1, S, Ko, blk, alpha=e, beta=1.0)
        del S
    if Kr.shape[0]:
        Vr = torch.empty(Kr.shape[0], Wg.shape[0], dtype=torch.float64, device=dev)
        hip.dgemm(0, 0, Kr, Wg, Vr)
        if bg is not None:
            Vr += bg
        hip.dgemm_ex(1, 1, Vr, Kr, blk, alpha=p_eff, beta=1.0)
    hip.cholesky_solve_(L, inv, rhs)
    out, r0 = [], 0
    for w in weights:
        out.append(rhs[r0:r0 + w.shape[0], :d].to(w.dtype).contiguous())
        r0 += w.shape[0]
    return out


def _pad_edits(K: torch.Tensor, seg: torch.Tensor, n_edits: int) -> torch.Tensor:
    """(M, d) rows grouped by edit (seg sorted) -> (n_edits, S, d), each edit's rows followed by zero rows."""
    lens = torch.bincount(seg, minlength=n_edits)
    S = int(lens.max().item())
    S += S % 2                                                      # even leading dimension for the f64 GEMM
    start = torch.cumsum(lens, 0) - lens
    pos = torch.arange(S, device=K.device)
    idx = torch.where(pos[None, :] < lens[:, None], start[:, None] + pos[None, :], K.shape[0])
    return torch.cat([K, torch.zeros(1, K.shape[1], dtype=K.dtype, device=K.device)], 0)[idx]


def _solve_shared(Ko, Kn, Kr, seg, n_edits, weights, L, inv, lamb, e, p_eff, technique):
    """Bias-free projections that share the context rows (the UNet variant), without ever forming a value row:
    S_l^T Ko = W_l (Kn^T Ko - sum_i alpha_li Ko_i^T Ko_i) with alpha_li = <W_l^T W_l, Ko_i^T Kn_i> / <W_l^T W_l, Ko_i^T Ko_i>
    (Frobenius products), so the per-edit d x d Grams are formed ONCE (batched GEMM) and every projection costs two
    contractions against them instead of two (M x d x out) GEMMs: ~25x fewer flops at SD-v1.4 shapes.
    mat1_l = W_l B_l,  B_l = lam I + e (Kn^T Ko - sum_i alpha_li G_i) + p Kr^T Kr;  W_new,l = mat1_l mat2^-1."""
    dev = Ko.device
    d = Ko.shape[1]
    dp = L.shape[0]
    P = len(weights)
    f64 = dict(dtype=torch.float64, device=dev)
    base = torch.zeros(d, d, **f64)
    base.diagonal().fill_(lamb)
    if Ko.shape[0]:
        hip.dgemm(1, 1, Kn, Ko, base, alpha=e, beta=1.0)             # e Kn^T Ko
    if Kr.shape[0]:
        hip.dgemm(1, 1, Kr, Kr, base, alpha=p_eff, beta=1.0)         # p Kr^T Kr
    W64 = [w.detach().double().contiguous() for w in weights]
    Bl = base.reshape(1, d * d).repeat(P, 1)                         # (P, d*d): B_l, row-major d x d each
    if technique == "tensor" and Ko.shape[0]:
        A = torch.empty(P, d * d, **f64)                             # W_l^T W_l
        for k, w in enumerate(W64):
            hip.dgemm(1, 1, w, w, A[k].view(d, d))
        Kop, Knp = _pad_edits(Ko, seg, n_edits), _pad_edits(Kn, seg, n_edits)
        for c0 in range(0, n_edits, EDIT_CHUNK):
            c1 = min(n_edits, c0 + EDIT_CHUNK)
            c = c1 - c0
            ce = c + c % 2                                           # even, the spare slot stays zero
            Goo = torch.zeros(ce, d, d, **f64)
            Gon = torch.zeros(ce, d, d, **f64)
            hip.dgemm_batched(1, 1, Kop[c0:c1], Kop[c0:c1], Goo[:c])
            hip.dgemm_batched(1, 1, Kop[c0:c1], Knp[c0:c1], Gon[:c])
            dot = torch.zeros(P, ce, **f64)
            sq = torch.zeros(P, ce, **f64)
            hip.dgemm(0, 0, A, Gon.view(ce, d * d), dot, beta=1.0)   # beta 1: the d*d-long contraction is split over workgroups
            hip.dgemm(0, 0, A, Goo.view(ce, d * d), sq, beta=1.0)
            alpha = torch.where(sq != 0, dot / sq, torch.zeros_like(dot))
            if c < ce:
                alpha[:, c:] = 0
            hip.dgemm(0, 1, alpha, Goo.view(ce, d * d), Bl, alpha=-e, beta=1.0)
            del Goo, Gon
    rhs = torch.zeros(sum(w.shape[0] for w in W64), dp, **f64)
    r0 = 0
    for k, w in enumerate(W64):
        hip.dgemm(0, 1, w, Bl[k].view(d, d), rhs[r0:r0 + w.shape[0], :d])
        r0 += w.shape[0]
    hip.cholesky_solve_(L, inv, rhs)
    out, r0 = [], 0
    for w in weights:
        out.append(rhs[r0:r0 + w.shape[0], :d].to(w.dtype).contiguous())
        r0 += w.shape[0]
    return out


def closed_form(Ko, Kn, Kr, seg, n_edits, weights, biases, lamb, erase_scale, preserve_scale, technique="tensor",
                method="auto"):
    """The whole device side for projections that share their context rows: Ko / Kn (M, d) f64 old / new rows, seg (M,)
    edit index per row (sorted), Kr (Mr, d) retain rows (context and value source alike), weights [(out_k, d)], biases
    [(out_k,) | None].  method: "rows" forms the value rows (any projection), "grams" contracts per-edit Grams
    (bias-free projections), "auto" = grams when no projection has a bias.  Returns the new weights, in the dtype of the
    old ones."""
    weights, biases = list(weights), list(biases)
    if method == "auto":
        method = "grams" if all(b is None for b in biases) else "rows"
    L, inv = _normal_matrix(Ko, Kr, float(lamb), float(erase_scale), float(preserve_scale))
    if method == "grams":
        if any(b is not None for b in biases):
            raise ValueError("method='grams' needs bias-free projections")
        return _solve_shared(Ko, Kn, Kr, seg, n_edits, weights, L, inv, float(lamb), float(erase_scale),
                             float(preserve_scale), technique)
    out = []
    for group in _column_groups(weights):
        out += _solve_group(Ko, Kn, Kr, seg, n_edits, [weights[k] for k in group], [biases[k] for k in group], L, inv,
                            float(lamb), float(erase_scale), float(preserve_scale), technique)
    return out


def _column_groups(weights) -> List[List[int]]:
    groups, group, cols = [], [], 0
    for k, w in enumerate(weights):
        if group and cols + w.shape[0] > MAX_VALUE_COLS:
            groups.append(group)
            group, cols = [], 0
        group.append(k)
        cols += w.shape[0]
    if group:
        groups.append(group)
    return groups


def _require_gpu(pipe):
    if torch.device(pipe.device).type != "cuda":
        raise hip.EmcidHipError("uce_train runs on the HIP path only: move the pipe to a GPU (no CPU fallback)")


# ---- entry points --------------------------------------------------------------------------------------------------

def edit_text_encoder_uce(pipe, old_text_, new_text_, retain_text_, add=False, layer_to_edit=11, lamb=0.1,
                          erase_scale=0.1, preserve_scale=0.1, with_to_k=True, technique='tensor'):
    """uce_train.py:31-213.  The reference's retain pass sits inside its loop over edits (:178), so the retain texts
    weigh `len(edits)` times `preserve_scale`; the values come from the whole fc2 module, bias included (:158); only
    the weight is replaced (:208).  `add` and `with_to_k` are accepted and unused, as there."""
    _require_gpu(pipe)
    import time
    t0 = time.perf_counter()
    module_name = f"text_model.encoder.layers.{layer_to_edit}.mlp.fc2"
    module = nethook.get_module(pipe.text_encoder, module_name)
    tap = next(n for n, m in pipe.text_encoder.named_modules() if m is module)
    old_texts, new_texts, ret_texts = _format_texts(old_text_, new_text_, retain_text_)
    n = len(old_texts)
    texts = [t for pair in zip(old_texts, new_texts) for t in pair] + ret_texts
    ti = _tokenize(pipe.tokenizer, texts)
    S = ti.input_ids.shape[1]
    old_flat, new_flat, seg = row_windows(ti.attention_mask.numpy(), n, S) if n else (np.zeros(0, np.int64),) * 3
    ret_flat = 2 * n * S + np.arange(len(ret_texts) * S)
    Ko, Kn, Kr = _encode_rows(pipe, ti.input_ids, (old_flat, new_flat, ret_flat), tap)
    if n == 0:
        return pipe                                  # the reference's loop body never runs: mat1 @ inv(mat2) = W
    seg_d = torch.from_numpy(seg).to(pipe.device)
    p_eff = float(preserve_scale) * n                # once per edit
    t1 = time.perf_counter()
    L, inv = _normal_matrix(Ko, Kr, float(lamb), float(erase_scale), p_eff)
    (new_w,) = _solve_group(Ko, Kn, Kr, seg_d, n, [module.weight], [module.bias], L, inv, float(lamb), float(erase_scale),
                            p_eff, technique)
    module.weight = torch.nn.Parameter(new_w)
    torch.cuda.synchronize()
    LAST_RUN.update(rows=int(Ko.shape[0]), retain_rows=int(Kr.shape[0]), forward_s=t1 - t0, solve_s=time.perf_counter() - t1)
    return pipe


def projection_entries(pipe, with_to_k: bool = True) -> Tuple[List[str], List[bool]]:
    """The reference's projection list and which of its entries reach the UNet (uce_train.py:232-260).  The list is
    built from the to_k/to_v layer names with the suffix stripped, so each attention block is in it twice: 2 x 16 `to_v`
    entries, then (with_to_k) 2 x 16 `to_k` entries; `layers_to_edit` indexes this doubled list.  The reference's reset
    loop leaves the first entry of each pair bound to a detached copy — editing it changes nothing — so only the last
    entry of a name is `attached`."""
    blocks = [n.replace('.to_k', '').replace('.to_v', '') for n in get_all_cross_attn_kv_layer_names(pipe)]
    nb = len(blocks)
    entries = [b + '.to_v' for b in blocks] + ([b + '.to_k' for b in blocks] if with_to_k else [])
    attached = []
    for j, name in enumerate(entries):
        half_end = nb * (j // nb + 1)
        attached.append(name not in entries[j + 1:half_end])
    return entries, attached


def edit_model_uce(ldm_stable, old_text_, new_text_, retain_text_, add=False, layers_to_edit=None, lamb=0.1,
                   erase_scale=0.1, preserve_scale=0.1, with_to_k=True, technique='tensor'):
    """uce_train.py:216-416: every cross-attention to_v (and to_k) of the UNet, each from its own original weight; the
    retain pass is outside the loop over edits here (:392), so it weighs `preserve_scale` once."""
    _require_gpu(ldm_stable)
    import time
    t0 = time.perf_counter()
    layers_to_edit = ast.literal_eval(layers_to_edit) if isinstance(layers_to_edit, str) else layers_to_edit
    lamb = ast.literal_eval(lamb) if isinstance(lamb, str) else lamb
    entries, attached = projection_entries(ldm_stable, with_to_k)
    wanted = [name for j, name in enumerate(entries)
              if attached[j] and (layers_to_edit is None or j in layers_to_edit)]
    old_texts, new_texts, ret_texts = _format_texts(old_text_, new_text_, retain_text_)
    n = len(old_texts)
    if not wanted:
        return ldm_stable
    texts = [t for pair in zip(old_texts, new_texts) for t in pair] + ret_texts
    ti = _tokenize(ldm_stable.tokenizer, texts)
    S = ti.input_ids.shape[1]
    old_flat, new_flat, seg = row_windows(ti.attention_mask.numpy(), n, S) if n else (np.zeros(0, np.int64),) * 3
    ret_flat = 2 * n * S + np.arange(len(ret_texts) * S)
    Ko, Kn, Kr = _encode_rows(ldm_stable, ti.input_ids, (old_flat, new_flat, ret_flat), None)
    seg_d = torch.from_numpy(seg).to(ldm_stable.device)
    t1 = time.perf_counter()
    mods = dict(ldm_stable.unet.named_modules())
    new_ws = closed_form(Ko, Kn, Kr, seg_d, n, [mods[g].weight for g in wanted], [mods[g].bias for g in wanted], lamb,
                         erase_scale, preserve_scale, technique, method=UCE_METHOD)
    for g, w in zip(wanted, new_ws):
        mods[g].weight = torch.nn.Parameter(w)
    torch.cuda.synchronize()
    LAST_RUN.update(rows=int(Ko.shape[0]), retain_rows=int(Kr.shape[0]), projections=len(wanted),
                    forward_s=t1 - t0, solve_s=time.perf_counter() - t1)
    return ldm_stable
