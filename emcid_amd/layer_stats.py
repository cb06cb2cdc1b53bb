"""Stage 0: per-layer second moment of the fc2 INPUT features over a caption set, npz-cached.

Host-side counterpart of the reference's emcid/layer_stats.py (``layer_stats_text_encoder`` :140-220,
``main`` :34-134).  Same sample (FixedRandomSubsetSampler seed 1), same groups of 100 captions, same
length-sorted <= batch_tokens sub-batches, same attended-token selection, same npz path and keys.

MI355X-first differences:
* the Gram accumulate is the hand-written SYRK kernel behind ``SecondMoment`` (csrc/gram_f32.hip);
* ``layer_stats_text_encoder_multi`` hooks EVERY requested layer in one forward (stopping after the
  deepest), so 12 layers cost 12 layer-forwards per batch instead of the reference's 78 (it re-runs the
  encoder from scratch for each layer, layer_stats.py:112-134);
* ``shard=(rank, world)`` partitions the fixed caption sample over ranks; the per-rank Grams are summed
  with one all-reduce per layer at the end (RCCL over xGMI on the GPU box).
"""
import argparse
import os
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import torch
from tqdm.auto import tqdm

from .clip_attention import hip_attention
from .globals import STATS_DIR, UNET_EDIT_TEMPLATES
from .nethook import StopForward, get_module, set_requires_grad
from .runningstats import (CombinedStat, SecondMoment, load_cached_state, make_loader, make_sampler,
                           save_cached_state, tally)
from .stat_dataset import TokenizedDataset, collate_token_lists, dict_to_, flatten_masked_batch, length_collation

STAT_TYPES = {"mom2": SecondMoment}
CCS_PATH = "./data/ccs_filtered.json"   # reference: layer_stats.py:138 (hard-coded, cwd-relative)


def get_ccs_filtered_ds(tokenizer, data_path=CCS_PATH):
    return TokenizedDataset(data_path, tokenizer)


def stats_filename(stats_dir, model_name, ds_name, layer_name, precision, to_collect, batch_tokens, sample_size) -> Path:
    # reference: layer_stats.py:166-174
    size_suffix = "" if sample_size is None else f"_{sample_size}"
    size_suffix = f"_t{batch_tokens}" + size_suffix
    return Path(stats_dir) / f"{model_name}/{ds_name}_stats/{layer_name}_{precision}_{'-'.join(sorted(to_collect))}{size_suffix}.npz"


def _check_precision(precision):
    if precision is None:
        precision = "float64"   # the reference's default when unset (layer_stats.py:161-162)
    if precision not in ("float16", "float32", "float64"):       # the reference CLI's choices (layer_stats.py:51)
        raise NotImplementedError(f"precision={precision!r}: float16, float32 (reference CLI default) or float64")
    return precision


def _stat_dtypes(precision):
    """(dtype the features are rounded to, dtype they are accumulated in, dtype the sums are stored in).  float32 / float64:
    all three that type (fp32 / fp64 MFMA SYRK).  float16: the reference casts the features to half and accumulates
    ``mom2 += a.t().mm(a)`` in half (layer_stats.py:51,:218; runningstats.py:493) — sums that saturate at 65 504 after a
    few thousand tokens; here the features are rounded to half exactly like that, the sums are carried in fp32 (no kernel
    accumulates in half) and only the stored matrix is half, so the file has the reference's schema and a value that is the
    half-rounded exact sum wherever the reference's is finite."""
    if precision == "float16":
        return torch.float16, torch.float32, torch.float16
    t = getattr(torch, precision)
    return t, t, t


def layer_stats_text_encoder_multi(model, tokenizer, layer_names: Sequence[str], stats_dir=STATS_DIR,
                                   ds_name="ccs_filtered", to_collect=("mom2",), model_name="text_encoder",
                                   sample_size=None, precision="float32", batch_tokens=3 * 1024, progress=tqdm,
                                   force_recompute=False, data_path=CCS_PATH, shard=None, group=None,
                                   num_workers=2, batch_size=100, device_batch_tokens=32768, feature: str = "input",
                                   files: Optional[Dict[str, Path]] = None, forward: str = "auto") -> Dict[str, CombinedStat]:
    """All ``layer_names`` in ONE pass over the captions.  Returns {layer_name: CombinedStat} (on cpu).

    ``batch_tokens`` names the cache file like the reference (``_t3072_``) but the device batches are larger:
    captions are pooled ``device_batch_tokens`` at a time, length-sorted and padded — the reference's 100-caption /
    3 072-token sub-batches keep a 12-layer forward at ~2 ms of GPU work under ~4 ms of Python.  The statistic is a
    sum over attended tokens, so the batch shape only changes the fp32 summation order."""
    precision = _check_precision(precision)
    to_collect = list(to_collect)
    device = next(model.parameters()).device
    args = {} if sample_size is None else {"sample_size": sample_size}
    # ``feature``: "input" (the reference's retain_input, layer_stats.py:209) or "output" of the hooked module;
    # ``files``: cache path per layer when it is not derived from the layer name (cross-attention statistics)
    files = files or {ln: stats_filename(stats_dir, model_name, ds_name, ln, precision, to_collect, batch_tokens,
                                         sample_size) for ln in layer_names}
    stats: Dict[str, CombinedStat] = {}
    todo: List[str] = []
    for ln in layer_names:
        st = CombinedStat(**{k: STAT_TYPES[k]() for k in to_collect})
        cached = None if force_recompute else load_cached_state(files[ln], args, quiet=True)
        if cached is not None:
            st.load_state_dict(cached)
        else:
            todo.append(ln)
        stats[ln] = st
    if not todo:
        return stats

    ds = get_ccs_filtered_ds(tokenizer, data_path)
    # Same fixed sample, same groups of ``batch_size`` captions, same length_collation as the reference's DataLoader
    # (layer_stats.py:196-206) — but the captions are tokenized with ONE batched tokenizer call up front instead of
    # one ``encode`` per item inside DataLoader workers (the pass was host-bound on that: 62k -> tokens/s).
    sample = list(make_sampler(ds, sample_size=sample_size, random_sample=1, shard=shard))
    pool = max(batch_size, (device_batch_tokens // 16) // batch_size * batch_size)   # captions pooled per collation

    def groups():
        # tokenized pool by pool, inside the loop: the GPU works on pool i (launches are asynchronous) while the host
        # tokenizes pool i+1 — tokenizing all captions up front kept the GPU idle for the first ~2.5 s of a 100k-caption job
        for g in range(0, len(sample), pool):
            ids = tokenize_ragged(tokenizer, [ds.data[i] for i in sample[g:g + pool]], ds.maxlen)
            yield collate_token_lists(ids, max(batch_tokens, device_batch_tokens))

    packed = _packed_plan(model, todo, mods_of=None) if (forward in ("auto", "trie") and feature == "input") else None
    if forward == "trie" and packed is None:
        raise ValueError("forward='trie' needs a HF CLIP text encoder and fc2 layer names")
    if packed is not None:
        return _collect_packed(model, tokenizer, ds, sample, pool, packed, todo, stats, files, args, shard, group, device,
                               progress, device_batch_tokens, _stat_dtypes(precision))
    loader = groups()
    round_dtype, stat_dtype, store_dtype = _stat_dtypes(precision)
    for ln in todo:
        if getattr(stats[ln], "mom2", None) is not None:
            stats[ln].mom2.store_dtype = store_dtype
    # forward order of the hooked modules decides which one is "deepest" (the one that stops the pass)
    order = {name: i for i, (name, _) in enumerate(model.named_modules())}
    mods = {ln: get_module(model, ln) for ln in todo}
    rank_of = {ln: next(order[n] for n, m in model.named_modules() if m is mods[ln]) for ln in todo}
    deepest = max(todo, key=lambda ln: rank_of[ln])
    grabbed: Dict[str, torch.Tensor] = {}
    handles = []
    for ln in todo:
        def hook(mod, inputs, output, ln=ln):
            grabbed[ln] = inputs[0] if feature == "input" else output
            if ln == deepest:
                raise StopForward()
        handles.append(mods[ln].register_forward_hook(hook))
    n_groups = -(-len(sample) // pool)
    wrap = progress if progress is not None else (lambda it, total=None: it)
    try:
        with torch.no_grad(), hip_attention(model):
            for batch_group in wrap(loader, total=n_groups):
                for batch in batch_group:
                    attended = batch.pop("attended").to(device, non_blocking=True)   # host-computed: no device nonzero
                    batch = dict_to_(batch, device)
                    try:
                        model(**batch)
                    except StopForward:
                        pass
                    for ln in todo:
                        x = grabbed[ln]
                        feats = x.reshape(-1, x.size(-1)).index_select(0, attended)  # == flatten_masked_batch(x, mask)
                        stats[ln].add(feats.to(dtype=round_dtype).to(dtype=stat_dtype))        # reference: feats.to(dtype=dtype) (:218)
                    grabbed.clear()
    finally:
        for h in handles:
            h.remove()
    _finish_layers(todo, stats, files, args, shard, group)
    return stats


def _finish_layers(todo, stats, files, args, shard, group):
    """all-reduce (caption-sharded runs), read-out and npz write of every collected layer"""
    for ln in todo:
        if shard is not None and shard[1] > 1:
            stats[ln].all_reduce_(group)
    # read-out and npz writes of the layers side by side (12 x 37.7 MB at SD dims: the device-to-host copies and the
    # uncompressed zip writes are memcpy / file-system time, serial they were ~10 % of a one-GPU Stage 0)
    def finish(ln):
        stats[ln].to_(device="cpu")
        if shard is None or shard[0] == 0:
            save_cached_state(files[ln], stats[ln], args)
    if len(todo) > 1:
        # Kernels stay on the CALLING thread and its stream: the last Gram flush and the symmetrize pass of every layer are
        # issued here, in order behind the accumulations (which may sit on a caller-chosen stream), and completed before the
        # workers start; the workers only copy finished matrices to the host and write files.
        cuda_devs = set()
        for ln in todo:
            m2 = getattr(stats[ln], "mom2", None)
            full = getattr(m2, "mom2", None) if m2 is not None else None
            if torch.is_tensor(full) and full.is_cuda:
                cuda_devs.add(full.device)
        for dv in cuda_devs:
            torch.cuda.current_stream(dv).synchronize()
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(4, len(todo))) as ex:
            list(ex.map(finish, todo))
    else:
        for ln in todo:
            finish(ln)


LAST_RUN = {}   # bookkeeping of the most recent Stage-0 pass (rows actually pushed through the Gram kernel vs tokens they stand for)


def _packed_plan(model, layer_names, mods_of=None):
    """(graph, {layer_name: encoder layer index}) when every requested module is the fc2 of a layer of a HF CLIP text
    model (then Stage 0 can run on the packed prefix trie), else None."""
    from . import clip_forward
    for tmpl in ("text_model.encoder.layers.{}", "encoder.layers.{}"):
        try:
            graph = clip_forward.discover(model, tmpl)
        except clip_forward.UnsupportedEncoder:
            continue
        index = {}
        for ln in layer_names:
            mod = get_module(model, ln)
            hit = [i for i, l in enumerate(graph.layers) if l.fc2 is mod]
            if not hit:
                return None
            index[ln] = hit[0]
        return graph, index
    return None


def tokenize_ragged(tokenizer, texts, max_length):
    """``tokenizer(texts, truncation=True, max_length=max_length)["input_ids"]`` (ragged lists, what the reference's
    TokenizedDataset feeds Stage 0: dsets/stat_dataset.py:71-103) without transformers' per-caption Python conversion of every
    encoding (1.0 of the 3.1 s of a 100 000-caption pass): a ONE-caption public call configures the backend's truncation exactly
    as transformers does for these arguments, then the backend encodes the batch without offsets; the longest caption is checked
    against the public call every time, anything unexpected takes the public call."""
    bt = getattr(tokenizer, "_tokenizer", None)
    if bt is not None and hasattr(bt, "encode_batch_fast") and len(texts) > 8:
        try:
            longest = max(range(len(texts)), key=lambda i: len(texts[i]))
            want = tokenizer([texts[longest]], truncation=True, max_length=max_length)["input_ids"][0]
            encs = bt.encode_batch_fast(list(texts), add_special_tokens=True)
            ids = [e.ids for e in encs]
            if len(ids) == len(texts) and ids[longest] == want and max(len(r) for r in ids) <= (max_length or 1 << 30):
                return ids
        except Exception:
            pass
    return tokenizer(list(texts), truncation=True, max_length=max_length)["input_ids"]


def _collect_packed(model, tokenizer, ds, sample, pool, packed, todo, stats, files, args, shard, group, device, progress,
                    device_batch_tokens, stat_dtypes=(torch.float32, torch.float32, torch.float32)):
    """Stage 0 on the packed prefix trie: the captions of a pool share their common prefixes ("<bos> a photo of ...") and
    carry no padding; every DISTINCT prefix is a row, computed once and entered into the Gram scaled by the square root
    of the number of captions that pass through it — sum_tokens x x^T exactly as the reference's attended-token sum
    (causal encoder: a token's state depends on its prefix only), in fp32 up to summation order.  Same fixed sample,
    same caption pools as the hooked forward."""
    from . import clip_forward
    graph, index = packed
    round_dtype, stat_dtype, store_dtype = stat_dtypes
    for ln in todo:
        if getattr(stats[ln], "mom2", None) is not None:
            stats[ln].mom2.store_dtype = store_dtype
    deepest = max(index.values())
    wanted = {i: [ln for ln in todo if index[ln] == i] for i in set(index.values())}
    wrap = progress if progress is not None else (lambda it, total=None: it)
    # larger pools than the hooked forward's: more shared prefixes per trie and longer GEMMs (rows ~ 0.8 x tokens);
    # 4 x the hooked pool per pool, i.e. ~110 k tokens, ~1.4 GB of fc2 inputs
    pool = 4 * pool
    pools = range(0, len(sample), pool)
    LAST_RUN.clear()
    LAST_RUN.update(forward="packed-trie", tokens=0, rows=0)
    with torch.no_grad():
        for g in wrap(pools, total=len(pools)):
            ids = tokenize_ragged(tokenizer, [ds.data[i] for i in sample[g:g + pool]], ds.maxlen)
            trie, cnt = clip_forward.build_trie_packed(ids, device)
            n_real, tokens = trie.n_nodes, int(sum(len(s) for s in ids))
            LAST_RUN["tokens"] += tokens
            LAST_RUN["rows"] += n_real
            root = cnt.to(stat_dtype).sqrt()

            def on_fc2(i, x, out):
                if i in wanted:
                    # the rows and their weights: the Gram's kernels form fl(root * row) as they read a long fp32 batch (one pass
                    # over ~1 GB per layer less than multiplying here); SecondMoment.add multiplies where that does not apply
                    feats = x[:n_real].to(round_dtype).to(stat_dtype)
                    for ln in wanted[i]:
                        stats[ln].add(feats, count=tokens, row_weight=root)
                return None if i == deepest else out

            clip_forward.run_layers(graph, trie, deepest, on_fc2, last_rows_only=False, fc2_by_callback={deepest})
    _finish_layers(todo, stats, files, args, shard, group)
    return stats


def layer_stats_text_encoder(model, tokenizer, layer_name, stats_dir="data/stats", ds_name="ccs_filtered",
                             to_collect=["mom2"], model_name="text_encoder", sample_size=None, precision=None,
                             batch_tokens=3 * 1024, download=False, progress=tqdm, force_recompute=False,
                             data_path=CCS_PATH, shard=None, group=None, num_workers=2):
    """Load or compute the cached stats of ONE layer (the reference's signature, layer_stats.py:140-153)."""
    if download:
        raise NotImplementedError("Downloading stats from remote is not implemented yet.")  # as the reference (:178)
    return layer_stats_text_encoder_multi(
        model, tokenizer, [layer_name], stats_dir, ds_name, to_collect, model_name, sample_size, precision,
        batch_tokens, progress, force_recompute, data_path, shard, group, num_workers)[layer_name]


# ---- cross-attention K/V statistics (reference: layer_stats.py:333-427, :429-468, :470-495, :555-575) ------------

def get_attr_through_name(obj, name):
    """Recursive getattr for dotted names (reference: layer_stats.py:577-579)."""
    for part in name.split("."):
        obj = getattr(obj, part)
    return obj


def get_to_edit_layername_unet(template_key, block_type, block_idx, sub_idx) -> str:
    """Module name of a UNet projection (reference: layer_stats.py:555-575, the attention templates)."""
    name = UNET_EDIT_TEMPLATES[template_key].format(block_type, block_idx, sub_idx)
    if "mid_block" in block_type:
        name = name.replace(f"mid_block.{block_idx}.", "mid_block.")
    return name


def get_all_cross_attn_kv_layer_names(pipe) -> List[str]:
    """Every ``attn2.to_k`` / ``to_v`` the UNet has, in the reference's order (layer_stats.py:470-495): down, up, mid;
    inside a block all to_k then all to_v."""
    names = []
    for block_type, count in (("down_blocks", 4), ("up_blocks", 4), ("mid_block", 1)):
        for idx in range(count):
            for key in ("cross-k", "cross-v"):
                for sub_idx in (0, 1, 2):
                    name = get_to_edit_layername_unet(key, block_type, idx, sub_idx)
                    try:
                        get_attr_through_name(pipe.unet, name)
                    except AttributeError:
                        continue
                    names.append(name)
    return names


def _final_norm_name(text_encoder) -> str:
    for name, mod in text_encoder.named_modules():
        if name.endswith("final_layer_norm"):
            return name
    raise LookupError("text encoder has no final_layer_norm")


def layer_stats_cross_attn_kv(pipe, layer_name, stats_dir="data/stats", ds_name="ccs_filtered", to_collect=["mom2"],
                              model_name="unet", sample_size=None, precision=None, batch_tokens=3 * 1024, download=False,
                              progress=tqdm, force_recompute=False, data_path=CCS_PATH, also: Sequence[str] = ()):
    """Load or compute the cached second moment of ``layer_name``'s input (reference: layer_stats.py:333-427).

    The input of every cross-attention K/V projection is the text encoder's last hidden state — the reference runs the
    UNet on dummy latents per caption batch only to hook it (:411-423).  Here the statistic is one pass of the text
    encoder (tap: output of its final LayerNorm, attended tokens) and is written under ``layer_name`` and every name in
    ``also`` (the reference computes all of them "in one go" too, :441-443), in the reference's path and npz format."""
    if download:
        raise NotImplementedError("Downloading stats from remote is not implemented yet.")   # as the reference (:368)
    precision = _check_precision(precision)
    to_collect = list(to_collect)
    args = {} if sample_size is None else {"sample_size": sample_size}
    names = [layer_name] + [n for n in also if n != layer_name]
    files = {n: stats_filename(stats_dir, model_name, ds_name, n, precision, to_collect, batch_tokens, sample_size)
             for n in names}
    if not force_recompute:
        cached = load_cached_state(files[layer_name], args, quiet=True)
        if cached is not None:
            st = CombinedStat(**{k: STAT_TYPES[k]() for k in to_collect})
            st.load_state_dict(cached)
            return st
    tap = _final_norm_name(pipe.text_encoder)
    scratch = files[layer_name].with_suffix(".tmp.npz")
    st = layer_stats_text_encoder_multi(pipe.text_encoder, pipe.tokenizer, [tap], stats_dir, ds_name, to_collect,
                                        model_name, sample_size, precision, batch_tokens, progress, True, data_path,
                                        feature="output", files={tap: scratch})[tap]
    scratch.unlink(missing_ok=True)
    for n in names:
        save_cached_state(files[n], st, args)
    return st


def compute_cross_attn_kv_stats(pipe, dataset="ccs_filtered", to_collect=["mom2"], sample_size=100000,
                                batch_tokens=3 * 1024, precision="float32", stats_dir=STATS_DIR, force_recompute=False,
                                data_path=CCS_PATH):
    """Statistics files of ALL cross-attention K/V projections from one text-encoder pass (reference: :429-468)."""
    names = get_all_cross_attn_kv_layer_names(pipe)
    return layer_stats_cross_attn_kv(pipe, names[0], stats_dir, dataset, to_collect, sample_size=sample_size,
                                     precision=precision, batch_tokens=batch_tokens, force_recompute=force_recompute,
                                     data_path=data_path, also=names[1:])


def main(argv=None):
    """Pre-compute cached stats for a text encoder held by the caller's pipeline loader.

    The reference's CLI (layer_stats.py:34-134) downloads Stable Diffusion from the hub; there is no network
    here, so this entry point runs on the synthetic encoders (``--model_name toy|sd-v1.4|sdxl-te1|sdxl-te2``)."""
    from . import synthetic as syn

    ap = argparse.ArgumentParser()
    ap.add_argument("--model_name", default="sd-v1.4", choices=list(syn.ENCODER_DIMS))
    ap.add_argument("--layers", default=12, type=int)
    ap.add_argument("--sample_size", default=100000, type=lambda x: None if x == "all" else int(x))
    ap.add_argument("--batch_tokens", default=3 * 1024, type=int)
    ap.add_argument("--precision", default="float32")
    ap.add_argument("--stats_dir", default=str(STATS_DIR))
    ap.add_argument("--data_path", default=CCS_PATH)
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args(argv)
    pipe = syn.build_pipe(a.model_name, a.device)
    set_requires_grad(False, pipe.text_encoder)
    names = [f"text_model.encoder.layers.{i}.mlp.fc2" for i in range(a.layers)]
    layer_stats_text_encoder_multi(pipe.text_encoder, pipe.tokenizer, names, a.stats_dir, sample_size=a.sample_size,
                                   precision=a.precision, batch_tokens=a.batch_tokens, data_path=a.data_path)


if __name__ == "__main__":
    main()
