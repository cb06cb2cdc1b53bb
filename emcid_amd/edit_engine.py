"""The restructured Stage-2 pass: ONE forward per encoder, the closed form solved inside the fc2 hooks.

The reference's layer loop (emcid/emcid_main.py:981-1073) runs the whole text encoder twice per edited layer
(once for the keys :987, once more for the current fc2 outputs :1004) on all N*P prompts — 98 % of its
wall-clock.  Layers are sequentially dependent: layer l+1's keys must see layer l's new
weights.  That dependency is exactly the order of a single forward pass, so here:

    forward(prompts)                                   # once
      └─ at each edited layer's fc2 (forward hook, in layer order):
           K  = gather_mean(fc2 input)                 # HIP, csrc/gram_f32.hip
           Zc = gather_mean(fc2 output)                # pre-edit output, bias included (as :1002-1016)
           [all-gather K, Zc over concept shards]      # RCCL, multi-GPU only
           W  = W0 + float(R X^T)                      # HIP fp64 MFMA solve, csrc/spd_solve.hip
           return fc2(input) with the NEW W            # the rest of the pass sees the edited layer
      └─ the hook of the last edited layer aborts the pass (nothing downstream is needed)

Same K, Zc, A, X, R, dW as the reference's loop (same inputs to every step), 1 partial forward instead of
2*L full ones.  Host work (tokenizing, subject search, v*/C loading) happens once in ``prepare``; ``run``
touches only HBM-resident inputs — that is the region bench.py times.
"""
import logging
import os
import threading
import time
import weakref
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence


import numpy as np
import torch
import torch.nn.functional as F

from . import hip
from . import clip_forward
from .clip_attention import hip_attention
from .compute_z import PromptBatch, build_prompt_batch, build_prompt_batch_multi, gather_request_means, iter_prompt_chunks
from .nethook import StopForward, get_module, get_parameter


@dataclass
class ConceptShard:
    """This rank's contiguous slice [lo, hi) of the request list (concept sharding)."""
    rank: int = 0
    world: int = 1
    group: object = None
    force_collectives: bool = False      # run the multi-rank code path (collectives, column-sharded solve) even at world size 1:
                                         # lets ONE GPU exercise the RCCL calls on HBM buffers (tests/test_dist_gpu.py)

    @property
    def collective(self) -> bool:
        return self.world > 1 or self.force_collectives

    def bounds(self, n: int, r: Optional[int] = None):
        r = self.rank if r is None else r
        return (n * r) // self.world, (n * (r + 1)) // self.world


@dataclass
class LayerEdit:
    layer: int
    weight_name: str
    dW: torch.Tensor                       # (h, d) fp32 in HBM: float(resid @ adj_k^T)
    Xt: Optional[torch.Tensor] = None      # (N, d) f64: adj_k^T
    Rt: Optional[torch.Tensor] = None      # (N, h) f64: resid^T
    K: Optional[torch.Tensor] = None       # (N, d) fp32 keys (kept when trace=True)
    Zc: Optional[torch.Tensor] = None      # (N, h) fp32 current fc2 outputs


@dataclass
class TrieChunk:
    """One slice of this rank's requests on the prefix-trie forward: its trie, its request -> prompt offsets and, when
    prepare already ran the unedited leading layers for it, the state that enters the first edited layer."""
    trie: clip_forward.TokenTrie
    seg: torch.Tensor                      # (n_requests + 1,) int64 prompt offsets
    n_requests: int
    n_prompts: int
    state: Optional[tuple] = None          # (layer index, residual stream, LN1 of it | None)


@dataclass
class EncoderEditPlan:
    """Everything one encoder's Stage-2 pass needs, resident in HBM."""
    text_encoder: torch.nn.Module
    layers: List[int]
    rewrite_module_tmp: str
    lam: float
    edit_weight: float
    batch: Optional[PromptBatch]           # this rank's prompts as one padded batch (hooked-HF forward; built on demand)
    zs_t: torch.Tensor                     # (N, h) fp32: v* of ALL requests, row per request
    covs: Dict[int, torch.Tensor]          # layer -> (d, d) fp32 second moment C
    n_total: int                           # N over all ranks
    shard: ConceptShard = field(default_factory=ConceptShard)
    ws: Optional[hip.EditWorkspace] = None
    dual_ws: Optional[hip.DualWorkspace] = None          # dual (Woodbury) solver state, see run_encoder_edit
    cov_factors: Optional[hip.CovFactors] = None
    side_stream: Optional[torch.cuda.Stream] = None
    solver: str = "auto"                                 # "direct" | "dual" | "auto" (dual when N is well below d)
    graph: Optional[clip_forward.ClipTextGraph] = None   # set -> prefix-deduplicated forward (clip_forward.py)
    chunks: Optional[List[TrieChunk]] = None             # the slices of the trie forward (None: hooked-HF forward)
    tokenizer: object = None
    local_requests: Optional[List[Dict]] = None
    zs_pending: object = None
    backups: Optional[Dict[int, torch.Tensor]] = None    # W0 of the edited layers of the last run (failure recovery)
    factor_key: Optional[tuple] = None   # set by a run that factored lam*C' itself: check_info caches the factors if sound
    factors_from_cache: bool = False
    num_edit_tokens: int = 1             # k > 1: k key / value rows per request (last subject token, EOS, padding), n_total = N k

    def weight_name(self, layer):
        return f"{self.rewrite_module_tmp.format(layer)}.weight"

    @property
    def trie(self):                        # the (first) trie of the prefix-deduplicated forward, None on the hooked-HF path
        return self.chunks[0].trie if self.chunks and self.graph is not None else None

    @trie.setter
    def trie(self, value):                 # ``plan.graph = plan.trie = None`` switches a plan to the hooked-HF forward
        if value is not None:
            raise AttributeError("assign plan.chunks instead")
        self.chunks = None

    @property
    def n_prompts(self) -> int:
        return sum(c.n_prompts for c in self.chunks) if self.chunks else self.ensure_batch().n_prompts

    @property
    def trie_rows(self):
        return (sum(c.trie.n_nodes for c in self.chunks), sum(c.trie.n_tokens_dense for c in self.chunks)) if self.chunks else None

    def resolve_targets(self) -> torch.Tensor:
        """(N, h) fp32 v* rows in HBM.  When prepare was handed the reader thread's future, this is where it is joined:
        at the first solve, i.e. after the forward up to the first edited layer has been launched."""
        if self.zs_pending is not None:
            with phase("vstar join + h2d"):
                with phase("vstar join (wait for the reader)"):
                    zs = self.zs_pending.result() if hasattr(self.zs_pending, "result") else self.zs_pending
                dev = next(self.text_encoder.parameters()).device
                if dev.type == "cuda" and not zs.is_cuda:
                    # through a page-locked staging buffer, asynchronously: a pageable copy would make the host wait for
                    # everything already queued on the stream (the encoder forward this call is meant to run underneath)
                    zs = _pinned_like(zs.to(torch.float32)).to(dev, non_blocking=True)
                else:
                    zs = zs.to(device=dev, dtype=torch.float32).contiguous()
                if zs.shape[0] != self.n_total:
                    raise ValueError(f"v* stack has {zs.shape[0]} rows for {self.n_total} requests")
                self.zs_t, self.zs_pending = zs, None
        return self.zs_t

    def ensure_batch(self) -> PromptBatch:
        if self.batch is None:
            dev = next(self.text_encoder.parameters()).device
            if self.num_edit_tokens > 1:
                self.batch = build_prompt_batch_multi(self.tokenizer, self.local_requests, dev, self.num_edit_tokens)
            else:
                self.batch = build_prompt_batch(self.tokenizer, self.local_requests, dev)
        return self.batch


# ---- state kept across calls (every access under ENGINE_LOCK) --------------------------------------------------------
ENGINE_LOCK = threading.RLock()
_WS_CACHE: "OrderedDict[tuple, list]" = OrderedDict()          # (kind, device, N, d, h) -> reusable f64 workspaces of that shape
_FACTOR_CACHE: "OrderedDict[tuple, tuple]" = OrderedDict()     # see factor_cache_key -> (CovFactors, cov tensors)
WS_CACHE_SIZE = 4
WS_PER_SHAPE = 2


def _factor_cache_size() -> int:
    return int(os.environ.get("EMCID_FACTOR_CACHE", "4"))


def _workspace(kind: str, N: int, d: int, h: int, dev, plan):
    """A workspace (buffers + its device `info` word) LEASED to ``plan`` until its check_info: two plans in flight —
    prepare(A), prepare(B), run(A), run(B), check_info(A) — or two threads never share buffers or a flag word; a plan that
    runs again gets its own workspace back.  A lease whose plan is gone (never checked) is free again."""
    key = (kind, str(dev), N, d, h)
    with ENGINE_LOCK:
        pool = _WS_CACHE.get(key)
        if pool is None:
            pool = _WS_CACHE[key] = []
            while len(_WS_CACHE) > WS_CACHE_SIZE:
                _WS_CACHE.popitem(last=False)
        else:
            _WS_CACHE.move_to_end(key)
        free = None
        for ws in pool:
            holder = ws.lease() if getattr(ws, "lease", None) is not None else None
            if holder is plan:
                return ws
            if holder is None and free is None:
                free = ws
        if free is None:
            free = {"dual": hip.DualWorkspace, "lu": hip.LuWorkspace}.get(kind, hip.EditWorkspace)(N, d, h, dev)
            if len(pool) < WS_PER_SHAPE:      # beyond that the workspace lives as long as its plan
                pool.append(free)
        free.lease = weakref.ref(plan)
        return free


def _release_workspaces(plan):
    with ENGINE_LOCK:
        for ws in (plan.ws, plan.dual_ws):
            if ws is not None and getattr(ws, "lease", None) is not None and ws.lease() is plan:
                ws.lease = None


def _pinned_like(t: torch.Tensor) -> torch.Tensor:
    """``t`` in page-locked memory: itself when it already is (the native v* reader fills pinned rows), else a copy in a
    buffer from torch's caching host allocator — which keeps a block alive until the asynchronous copies issued from it have
    finished (it records an event per stream use), so two plans prepared back to back never share a staging buffer."""
    if t.is_pinned():
        return t
    buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    buf.copy_(t)
    return buf


def factor_cache_key(covs: Sequence[torch.Tensor], lam: float, edit_weight: float) -> tuple:
    """The Cholesky factor of lam*C'_l and its explicit inverse are functions of (the statistics, edit_weight) up to a scalar:
    chol(lam C') = sqrt(lam) chol(C'), so ``lam`` is NOT part of the key — an edit with another mom2_update_weight (the
    reference's most common sweep, experiments/emcid_test.py:924-930, ablation.py:80-83) reuses the factors and hands the
    library the ratio (include/emcid_hip.h, "lam_ratio").  edit_weight stays in the key: C' = fl32(fl32(C (1 - e_w)) / 0.5)
    is rounded per entry in fp32 (reference :1037), which a scalar cannot reproduce.  The statistics are identified by the
    identity and version counter of the HBM-resident C tensors (the entries of emcid_main's covariance cache).
    Reusing a factor across lam changes the weights at fp64-rounding level against a process that factors lam C' itself;
    (A process that wants bit-stable output across runs keeps lam fixed.)"""
    key_lam = None
    key_ew = float(edit_weight)
    return (tuple((c.device.index, c.data_ptr(), c._version, tuple(c.shape)) for c in covs), key_ew, key_lam)


def solve_lam(plan) -> float:
    """The lam handed to the dual stages.  (The cached factors are keyed by the statistics AND edit_weight, so they always belong
    to this call's edit_weight; round 4's EMCID_EDIT_WEIGHT_SCALAR switch — reuse them across edit_weights by treating C' as a
    scalar multiple of C — left the 1e-4 bar on statistics of condition > 1e5 and is gone.)"""
    return plan.lam


def clear_engine_caches():
    with ENGINE_LOCK:
        _WS_CACHE.clear()
        _FACTOR_CACHE.clear()


TIMING: Dict[str, float] = {}      # host seconds per phase of the last prepare/run (diagnostic; scripts/host_profile.py)


class phase:
    """with phase("name"): ... accumulates host wall-clock into TIMING[name]."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.t = time.perf_counter()

    def __exit__(self, *exc):
        TIMING[self.name] = TIMING.get(self.name, 0.0) + time.perf_counter() - self.t
        return False


FORWARD_MODE = "trie"   # "trie": prefix-deduplicated forward when the encoder is a HF CLIP text model; "hf": hooked HF forward
SOLVER = None           # None: plan.solver decides; "direct" / "dual" force it (tests, experiments; env EMCID_SOLVER too)


def _solver_mode(plan) -> str:
    if plan.solver == "lu":          # the fallback after a failed Cholesky overrides every preference
        return "lu"
    return SOLVER or os.environ.get("EMCID_SOLVER") or plan.solver


def _use_dual(plan, d: int) -> bool:
    mode = _solver_mode(plan)
    if mode == "dual":
        return True
    if mode in ("direct", "lu"):
        return False
    np_, dp = -(-plan.n_total // hip.NB) * hip.NB, -(-d // hip.NB) * hip.NB
    return np_ * 5 <= dp * 3      # N x N system + two extra solves pay off when N is well below d


def prepare_encoder_edit(text_encoder, tokenizer, requests: Sequence[Dict], layers, rewrite_module_tmp, lam,
                         edit_weight, zs_t: torch.Tensor, covs: Dict[int, torch.Tensor],
                         shard: Optional[ConceptShard] = None, layer_module_tmp: Optional[str] = None,
                         forward_mode: Optional[str] = None, num_edit_tokens: int = 1, _defer_checks: bool = True) -> EncoderEditPlan:
    from . import manage_threads
    manage_threads()                    # acts once per process, and only under EMCID_MANAGE_THREADS=1
    shard = shard or ConceptShard()
    device = next(text_encoder.parameters()).device
    lo, hi = shard.bounds(len(requests))
    if len(requests) < shard.world:     # the same verdict on every rank, before any collective can be entered
        raise ValueError(f"{len(requests)} request(s) cannot be sharded over {shard.world} ranks (every rank needs one)")
    local = list(requests[lo:hi])
    mode = forward_mode or FORWARD_MODE
    graph = None
    k = int(num_edit_tokens)
    if k < 1:
        raise ValueError(f"num_edit_tokens must be >= 1, got {k}")
    if k > 1:
        # k rows per request ([last subject token, EOS, k - 2 padding positions], reference compute_z.py:2329-2382 and
        # emcid_main.py:993-1014): the rows behind the EOS need the padded prompts, so this runs on the hooked forward over the
        # padded batch (no prefix trie, no token truncation) and on one rank; the closed form then sees N k concepts
        if shard.collective:
            raise NotImplementedError("num_edit_tokens > 1 is a single-rank path")
        mode = "hf"
    if mode == "trie" and layer_module_tmp is not None and device.type == "cuda":
        try:
            with phase("graph"):
                graph = clip_forward.discover_cached(text_encoder, layer_module_tmp)
                for l in layers:   # the edited weights must be the very tensors the explicit forward multiplies with
                    if graph.layers[l].fc2.weight is not get_parameter(text_encoder, f"{rewrite_module_tmp.format(l)}.weight"):
                        raise clip_forward.UnsupportedEncoder("rewrite_module_tmp is not the layer's mlp.fc2")
                if sorted(layers) != list(layers):
                    raise clip_forward.UnsupportedEncoder("layers not in forward order")
        except (clip_forward.UnsupportedEncoder, IndexError, LookupError) as e:
            clip_forward.note_fallback("prepare_encoder_edit", e)
            graph = None
    plan = EncoderEditPlan(text_encoder, list(layers), rewrite_module_tmp, float(lam), float(edit_weight), None,
                           None, covs, len(requests) * k, shard, tokenizer=tokenizer, local_requests=local, num_edit_tokens=k)
    if graph is not None:
        # As soon as the prompts are tokenized their prefix trie is built and the unedited leading layers are LAUNCHED here,
        # so the GPU runs them underneath the rest of the host preparation (v* reads, statistics lookups).  The prompt list
        # could also be cut into slices (iter_prompt_chunks takes a count), each launched before the next is tokenized; measured on 2 x EPYC
        # 9575F (scripts/host_profile.py, configurations interleaved) every extra tokenizer call costs 3-4 ms of fixed
        # overhead (waking the backend's thread pool), more than the 2.9 ms of GPU time a second slice hides: default 1.
        n_chunks = 1
        first_edit = plan.layers[0]
        chunks: List[TrieChunk] = []
        try:
            it = iter_prompt_chunks(tokenizer, local, n_chunks, defer_probe=_defer_checks)
            while True:
                with phase("tokenize+lookup"):
                    pc = next(it, None)
                if pc is None:
                    break
                with phase("trie"):
                    trie = clip_forward.build_trie(pc.ids, pc.lookup, device, tail=pc.request_offsets())
                    seg = trie.tail
                with phase("prefix launches"), torch.no_grad():
                    hs, x_ln1 = clip_forward.run_prefix(graph, trie, first_edit)
                chunks.append(TrieChunk(trie, seg, pc.n_requests, len(pc.lookup), (first_edit, hs, x_ln1)))
                if pc.verify is not None:
                    # the checks of the templated tokenization that need not hold the first launch back — the native tokenizer's
                    # cross-check against the public tokenizer call, the reference's subject walk against the lookup positions
                    # known from the construction of the rows — run now that the GPU has this slice's leading layers to work on.
                    # If one says no (a name that also occurs earlier in its prompt; a tokenizer disagreement retires the native
                    # twin for good), the preparation starts over with every check up front; what was launched is abandoned.
                    with phase("tokenize+lookup"):
                        agreed = pc.verify()
                    if not agreed:
                        return prepare_encoder_edit(text_encoder, tokenizer, requests, layers, rewrite_module_tmp, lam, edit_weight,
                                                    zs_t, covs, shard, layer_module_tmp, forward_mode, num_edit_tokens,
                                                    _defer_checks=False)
            plan.graph, plan.chunks = graph, chunks
            # gauges (not counters): the trie of the last prepared call — rows run per layer and the tokens they stand for
            clip_forward.LAST_PATHS["last_trie_rows"], clip_forward.LAST_PATHS["last_trie_tokens"] = plan.trie_rows
        except (clip_forward.UnsupportedEncoder, IndexError) as e:
            clip_forward.note_fallback("prepare_encoder_edit (prompt batch)", e)
            plan.graph = plan.chunks = None
    if plan.chunks is None:
        with phase("tokenize+lookup"):
            plan.ensure_batch()
    # ``zs_t`` / ``covs`` may be callables: the caller's v* cache check and statistics lookups, run HERE — after the leading
    # layers have been launched, so that these host milliseconds too pass underneath the GPU (v* first, as the reference)
    if callable(zs_t) and not hasattr(zs_t, "result"):
        with phase("vstar check"):
            zs_t = zs_t()
    if callable(covs):
        with phase("statistics"):
            covs = covs()
    plan.covs = {l: c.to(device=device, dtype=torch.float32).contiguous() for l, c in covs.items()}
    plan.zs_pending = zs_t          # a tensor, or the caller's reader-thread future: resolved where the first solve needs it
    if not hasattr(zs_t, "result"):
        plan.resolve_targets()
    return plan


def _staged(group) -> bool:
    """True when the process group cannot take HBM tensors (gloo): collectives are then staged through host
    memory.  RCCL ("nccl") — the production backend — works on the device buffers directly."""
    import torch.distributed as dist
    return dist.get_backend(group) == "gloo"


# Per-rank phase times of a sharded edit (bench.py --gpus N reports them for every rank, so that a bad scaling curve can be read):
# EMCID_DIST_TIMING=1 (or DIST_TIMING["enabled"] = True) brackets the collectives and every layer's solve with event pairs on
# the launch stream; dist_timing_collect() synchronises and returns {phase: (total ms, count)}.
DIST_TIMING = {"enabled": os.environ.get("EMCID_DIST_TIMING", "0") == "1", "events": []}


class _dist_phase:
    def __init__(self, name: str, dev):
        self.name, self.dev = name, dev
        self.on = DIST_TIMING["enabled"] and torch.device(dev).type == "cuda"

    def __enter__(self):
        if self.on:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record(torch.cuda.current_stream(self.dev))
        return self

    def __exit__(self, *exc):
        if self.on:
            b = torch.cuda.Event(enable_timing=True)
            b.record(torch.cuda.current_stream(self.dev))
            DIST_TIMING["events"].append((self.name, self.a, b))
        return False


def dist_timing_collect() -> Dict[str, tuple]:
    """{phase: (milliseconds in total, number of brackets)} since the last collect (synchronises the device)."""
    out: Dict[str, list] = {}
    evs, DIST_TIMING["events"] = DIST_TIMING["events"], []
    if evs:
        torch.cuda.synchronize()
    for name, a, b in evs:
        rec = out.setdefault(name, [0.0, 0])
        rec[0] += a.elapsed_time(b)
        rec[1] += 1
    return {k: (v[0], v[1]) for k, v in out.items()}


def _all_reduce_sum(t: torch.Tensor, group):
    import torch.distributed as dist
    # the two all-reduces of the column-sharded solve: the N x N partial S (square) and the h x d partial U
    name = "all_reduce_S" if t.dim() == 2 and t.shape[0] == t.shape[1] else "all_reduce_U"
    with _dist_phase(name, t.device):
        if t.is_cuda and _staged(group):
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            t.copy_(host)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def _all_gather_rows(local: torch.Tensor, plan: EncoderEditPlan) -> torch.Tensor:
    """Concatenate per-rank row blocks (uneven shards allowed) in rank order == request order."""
    sh = plan.shard
    if not sh.collective:
        return local
    import torch.distributed as dist

    with _dist_phase("k_all_gather", local.device):
        return _all_gather_rows_timed(local, plan, sh, dist)


def _all_gather_rows_timed(local, plan, sh, dist):
    sizes = [b - a for a, b in (sh.bounds(plan.n_total, r) for r in range(sh.world))]
    nmax = max(sizes)
    padded = local
    if local.shape[0] < nmax:
        padded = torch.zeros(nmax, local.shape[1], dtype=local.dtype, device=local.device)
        padded[:local.shape[0]] = local
    out = torch.empty(sh.world * nmax, local.shape[1], dtype=local.dtype, device=local.device)
    if local.is_cuda and _staged(sh.group):
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, padded.cpu().contiguous(), group=sh.group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, padded.contiguous(), group=sh.group)
    if all(s == nmax for s in sizes):
        return out
    return torch.cat([out[r * nmax:r * nmax + s] for r, s in enumerate(sizes)], dim=0)


def _touch(w: torch.Tensor):
    """The kernels write an edited weight through its raw pointer, which torch's in-place version counter does not see; caches
    keyed by (tensor, version) — the split-fp16 copies of the weights, clip_forward.ClipLayer.split_of — must."""
    torch.autograd.graph.increment_version(w)


def run_encoder_edit(plan: EncoderEditPlan, keep_factors: bool = False, trace: bool = False,
                     restore: bool = False) -> List[LayerEdit]:
    """Device-only Stage 2 for one encoder.  On return the edited fc2 weights hold W0 + dW (``restore=False``)
    or their original values (``restore=True``, the reference's execute_* invariant, emcid_main.py:1076-1078)."""
    te = plan.text_encoder
    L = len(plan.layers)
    mods = {l: get_module(te, plan.rewrite_module_tmp.format(l)) for l in plan.layers}
    weights = {l: get_parameter(te, plan.weight_name(l)) for l in plan.layers}
    for l, w in weights.items():
        if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()):
            raise hip.EmcidHipError(f"{plan.weight_name(l)} must be a contiguous fp32 tensor in HBM "
                                    f"(got {w.dtype} on {w.device})")
    backups = {l: w.detach().clone() for l, w in weights.items()}
    plan.backups = backups
    guard = plan.graph.guard if plan.graph is not None else None
    if guard is not None:
        # the caches this pass runs on (stacked q | k | v, split planes, layer structs) against the BYTES of the live weights:
        # one launch, read back by check_info with the solver's flags (clip_forward.WeightGuard)
        guard.check(plan.graph.layers, plan.layers[-1] + 1)
    d, h = weights[plan.layers[0]].shape[1], weights[plan.layers[0]].shape[0]
    dev = weights[plan.layers[0]].device
    dual = _use_dual(plan, d)
    edits: List[LayerEdit] = []
    last = plan.layers[-1]
    handles = []
    fac_done = None
    lazy, first_x = False, 0
    plan.factor_key, plan.factors_from_cache = None, False
    if dual:
        plan.dual_ws = _workspace("dual", plan.n_total, d, h, dev, plan)
        plan.dual_ws.info.zero_()
        fkey = factor_cache_key([plan.covs[l] for l in plan.layers], plan.lam, plan.edit_weight)
        with ENGINE_LOCK:
            hit = _FACTOR_CACHE.get(fkey) if _factor_cache_size() > 0 and plan.lam > 0 else None
            if hit is not None and not (hit[0].lam and hit[0].lam > 0):
                hit = None
            if hit is not None:
                _FACTOR_CACHE.move_to_end(fkey)
        if hit is not None:
            # lam0 * C'_l = L L^T and X = inv(L) of every edited layer are already in HBM from an earlier edit with the
            # same statistics and edit_weight (any lam0: the solve stages take lam / lam0): every M-solve of this pass is a
            # GEMM against X, nothing is factored but the N x N systems
            plan.cov_factors, plan.factors_from_cache = hit[0], True
            if plan.cov_factors.ready is not None:
                torch.cuda.current_stream(dev).wait_event(plan.cov_factors.ready)
        else:
            # lam * C'_l does not depend on the concepts: factor it for ALL edited layers in one batched pass on a side
            # stream, underneath the encoder forward that produces the first layer's keys
            if plan.side_stream is None:
                # high priority: the factorization is a long chain of small dependent kernels; each must get the next
                # free CU ahead of the forward's wide GEMMs or the chain stretches to several times its own length
                plan.side_stream = torch.cuda.Stream(device=dev, priority=-1)
            plan.side_stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(plan.side_stream):
                if plan.cov_factors is not None and plan.cov_factors.cached:
                    plan.cov_factors = None        # never refactor into a workspace the cache hands to other edits
                if plan.cov_factors is not None:
                    plan.cov_factors.info.zero_()
                plan.cov_factors = hip.factor_cov([plan.covs[l] for l in plan.layers], plan.lam, plan.edit_weight,
                                                  plan.cov_factors, inverse=False)
                plan.factor_key = fkey
                chol_done = torch.cuda.Event()
                chol_done.record(plan.side_stream)
                # The first edited layer solves against M by block substitution with L as soon as the factorization is there.
                # The explicit inverse factors X_l = inv(L_l) of the LATER layers (their two M-solves become two GEMMs) are
                # built one layer ahead, on the side stream, exactly while the previous layer's solve sits in the
                # latency-bound Cholesky of its N x N system (the chip is idle there): see ``lazy_inverse`` in solve().
                # (Building them all right after the factorization, batched underneath the forward, cost the forward more than it
                # hid: HISTORY.md §5.)
                first_x = min(L, 1)
                lazy = keep_factors is False
                if plan.shard.collective and not keep_factors and -(-d // hip.NB) >= plan.shard.world:
                    first_x, lazy = 0, False      # the column-sharded solve multiplies by X in every layer
                fac_done = [chol_done] * L
                if lazy and first_x == 0:      # the first layer's X right behind the factorization, the others one layer ahead
                    hip.cov_inverse(plan.cov_factors, 0, 1)
                    fac_done[0] = torch.cuda.Event()
                    fac_done[0].record(plan.side_stream)
                if first_x < L and not lazy:
                    hip.cov_inverse(plan.cov_factors, first_x, L - first_x)
                    ev = torch.cuda.Event()
                    ev.record(plan.side_stream)
                    for i in range(first_x, L):
                        fac_done[i] = ev
            if first_x >= L:
                lazy = False
    elif _solver_mode(plan) == "lu":
        plan.ws = _workspace("lu", plan.n_total, d, h, dev, plan)
        plan.ws.info.zero_()
        plan.dual_ws = None
    else:
        plan.ws = _workspace("direct", plan.n_total, d, h, dev, plan)
        plan.ws.info.zero_()

    def solve(i, layer, K_local, Zc_local):
        """All-gather the shard's K/Zc rows, run the closed form, leave W0 + dW in the live weight.  ``Zc_local`` may be
        a callable K -> Zc (fc2 applied to the gathered keys): then only K crosses the links."""
        try:
            with _dist_phase("solve (incl. its collectives)", dev):
                _solve(i, layer, K_local, Zc_local)
        finally:
            _touch(weights[layer])

    def _solve(i, layer, K_local, Zc_local):
        K = _all_gather_rows(K_local, plan)
        Zc = Zc_local(K) if callable(Zc_local) else _all_gather_rows(Zc_local, plan)
        plan.resolve_targets()
        if dual:
            if fac_done is not None:
                torch.cuda.current_stream(dev).wait_event(fac_done[i])
            sharded = plan.shard.collective
            if not keep_factors:   # only the edited weights are wanted: the form that never builds adj_k
                def lazy_inverse(nxt=i + 1):
                    # stream position: S of layer i is assembled, its Cholesky starts now -> build X of the next layer
                    start = torch.cuda.Event()
                    start.record(torch.cuda.current_stream(dev))
                    plan.side_stream.wait_event(start)
                    with torch.cuda.stream(plan.side_stream):
                        hip.cov_inverse(plan.cov_factors, nxt, 1)
                        fac_done[nxt] = torch.cuda.Event()
                        fac_done[nxt].record(plan.side_stream)

                ahead = lazy and max(first_x, 1) <= i + 1 < L
                if sharded and -(-d // hip.NB) >= plan.shard.world:
                    # Every rank holds all N key rows (the K all-gather above).  The layer's GEMMs are split by 128-wide
                    # column tiles of d: a rank forms its columns of Yt = Kt X^T and its share of S = I + Yt Yt^T and of
                    # U = (Z^T Yt) X; S (N x N) and U (h x d) are summed over the ranks (two all-reduces over xGMI), the
                    # N x N Cholesky and the h-row solves are repeated by everyone (latency-bound, no d^2 work in them).
                    n_tiles = -(-d // hip.NB)
                    res = hip.edit_layer_dual_cols(
                        K, Zc, plan.zs_t, plan.cov_factors, i, plan.edit_weight, L - i, backups[layer], weights[layer].data,
                        hip.column_tiles(plan.shard.rank, plan.shard.world, n_tiles),
                        lambda t: _all_reduce_sum(t, plan.shard.group), ws=plan.dual_ws, lam=solve_lam(plan))
                    edits.append(LayerEdit(layer, plan.weight_name(layer), res["dW"], None, None,
                                           K if trace else None, Zc if trace else None))
                    return
                # (fewer 128-column tiles than ranks: every rank runs the whole layer itself)
                split = False
                res = hip.edit_layer_dual_apply(
                    K, Zc, plan.zs_t, plan.cov_factors, i, plan.edit_weight, L - i, backups[layer], weights[layer].data,
                    ws=plan.dual_ws, rows=plan.shard.bounds(plan.n_total) if split else None,
                    gather_yt=(lambda rows_: _all_gather_rows(rows_.contiguous(), plan)) if split else None,
                    on_factor_start=lazy_inverse if ahead else None, lam=solve_lam(plan))
                edits.append(LayerEdit(layer, plan.weight_name(layer), res["dW"], None, None,
                                       K if trace else None, Zc if trace else None))
                return
            res = hip.edit_layer_dual(
                K, Zc, plan.zs_t, plan.cov_factors, i, plan.edit_weight, L - i, W0=backups[layer], W=weights[layer].data,
                want_factors=keep_factors, ws=plan.dual_ws,
                rows=plan.shard.bounds(plan.n_total) if sharded else None,
                gather_pt=(lambda rows_: _all_gather_rows(rows_.contiguous(), plan)) if sharded else None, lam=solve_lam(plan))
            xt = res["adj_k"].t() if res["adj_k"] is not None else None
            edits.append(LayerEdit(layer, plan.weight_name(layer), res["dW"], xt, res["Rt"],
                                   K if trace else None, Zc if trace else None))
            return
        if _solver_mode(plan) == "lu":
            # the reference's own algorithm (LU with partial pivoting) for a system the Cholesky paths rejected; every
            # rank holds all N key rows here and solves the whole layer itself (rare path, never sharded)
            res = hip.edit_layer_lu(K, Zc, plan.zs_t, plan.covs[layer], plan.lam, plan.edit_weight, L - i,
                                    W0=backups[layer], W=weights[layer].data, want_factors=keep_factors, ws=plan.ws)
            xt = res["adj_k"].t() if res["adj_k"] is not None else None
            edits.append(LayerEdit(layer, plan.weight_name(layer), res["dW"], xt, res["Rt"],
                                   K if trace else None, Zc if trace else None))
            return
        if plan.shard.collective and not keep_factors:
            # every rank assembles and factors A from all N concepts; the triangular solves and the dW
            # contraction are split by concept rows and the partial U summed over xGMI (fp64, h*d*8 bytes)
            res = hip.edit_layer_shard(K, Zc, plan.zs_t, plan.covs[layer], plan.lam, plan.edit_weight, L - i,
                                       plan.shard.bounds(plan.n_total), ws=plan.ws)
            _all_reduce_sum(res["U"], plan.shard.group)
            dW = hip.apply_update_(res["U"], backups[layer], weights[layer].data)
            res = {"dW": dW, "Xt": None, "Rt": None}
        else:
            res = hip.edit_layer(K, Zc, plan.zs_t, plan.covs[layer], plan.lam, plan.edit_weight, L - i,
                                 W0=backups[layer], W=weights[layer].data, want_factors=keep_factors, ws=plan.ws)
        edits.append(LayerEdit(layer, plan.weight_name(layer), res["dW"], res["Xt"], res["Rt"],
                               K if trace else None, Zc if trace else None))

    if plan.chunks is not None and plan.graph is not None:
        chunks, order = plan.chunks, {l: i for i, l in enumerate(plan.layers)}
        if sorted(plan.layers) != plan.layers:
            raise RuntimeError("hparams.layers must be in forward order")
        first_edit = plan.layers[0]

        def rows_at(x, idx, ch):   # (rows, c) activations -> per-request means at each prompt's lookup row
            if isinstance(x, hip.SplitRows):       # split-fp16 path: the keys come from the fp32 twin fc1's epilogue wrote
                x = x.float()
            return hip.gather_mean(x.unsqueeze(0).expand(idx.numel(), -1, -1), idx, ch.seg)

        def keys(li, xs):          # this rank's (N_local, d) key rows, slices concatenated in request order
            parts = [rows_at(x, ch.trie.lookup_in_query if li == last else ch.trie.lookup_node, ch) for x, ch in zip(xs, chunks)]
            return parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)

        zc_from_keys = True      # Zc = fc2(mean keys): fc2 is affine (the other form, fc2 over every node + gather, left in round 5)
        fused_env = os.environ.get("EMCID_FUSED_EDIT_LAYER", "1") != "0"

        def fused_layer_ok(li, xs, mids):
            """The one-call form of an edited layer: single rank, one prompt slice, the apply-only dual solver on factors that
            are already in HBM (a warm call), the split-fp16 forward with its native runner."""
            if not (fused_env and zc_from_keys and dual and not keep_factors and not plan.shard.collective and mids is not None
                    and len(chunks) == 1 and plan.factors_from_cache and fac_done is None and clip_forward.NATIVE_RUNNER):
                return False
            x = xs[0]
            if not (isinstance(x, hip.SplitRows) and x.f32 is not None and x.f32.is_contiguous()):
                return False
            gl = plan.graph.layers[li]
            if gl.fc2 is not mods[li] or gl.fc2.bias is None:
                return False
            return clip_forward.native_of(plan.graph, chunks[0].trie, li, li + 1) is not None

        def on_fc2(li, xs, outs, mids=None, next_ln=None):
            """fc2 of an edited layer: keys -> closed form -> the projection with the NEW weight.  ``mids`` (the residual
            streams): the callback returns fc2(x) + mid, the add riding in the GEMM's epilogue; with the native layer runner
            (hs, LN1 planes of the next layer) pairs from ONE C call (``next_ln``: that LayerNorm, or None)."""
            if li not in order:
                return outs
            m = mods[li]
            if fused_layer_ok(li, xs, mids):
                # keys, Zc, the solve, the new weight's planes and fc2 + residual + the next LN1: ONE C call
                # (csrc/clip_layers.hip: emcid_clip_edit_layer_tail_sp16)
                ch, x, i = chunks[0], xs[0], order[li]
                nat = clip_forward.native_of(plan.graph, ch.trie, li, li + 1)
                plan.resolve_targets()
                w = weights[li]
                try:
                    res = hip.clip_edit_layer_tail(
                        nat.array, li, nat.h, nat.d, x, mids[0], ch.trie.lookup_in_query if li == last else ch.trie.lookup_node,
                        ch.seg, plan.zs_t, plan.cov_factors, i, plan.edit_weight, L - i, backups[li], w.data, plan.dual_ws,
                        solve_lam(plan), next_ln, last=li == last)
                finally:
                    _touch(w)
                if li != last:      # the layer's fc2 planes were re-split in place from the new weight: keep the cache entry
                    gl = plan.graph.layers[li]
                    gl.splits["fc2"] = ((id(gl.fc2.weight), gl.fc2.weight._version, gl.fc2.weight.data_ptr()), gl.splits["fc2"][1])
                    if gl.guard is not None:
                        gl.guard.store(gl.index, "fc2", gl.fc2.weight)
                edits.append(LayerEdit(li, plan.weight_name(li), res["dW"], None, None, res["K"] if trace else None,
                                       res["Zc"] if trace else None))
                clip_forward.LAST_PATHS["fused_edit_layers"] = clip_forward.LAST_PATHS.get("fused_edit_layers", 0) + 1
                return None if li == last else [(res["hs"], res["x"])]
            K_loc = keys(li, xs)
            lin = clip_forward.linear
            if not zc_from_keys:
                solve(order[li], li, K_loc, keys(li, [lin(x, m.weight, m.bias) for x in xs]))
            else:
                # fc2 is affine, so the mean over a request's prompts of its output at the lookup rows IS fc2 of the mean
                # key: Zc = K W^T + b on N rows (to fp32 rounding) instead of fc2 over every node followed by a gather
                # (on the split-fp16 kernel against the planes of the weight as it is now, like the fused call above)
                solve(order[li], li, K_loc, lambda K_all: lin(K_all, m.weight, m.bias, wsp=plan.graph.layers[li].split_of("fc2")))
            if li == last:
                return None
            wsp = plan.graph.layers[li].split_of("fc2")          # of the NEW weight (solve() bumped its version counter)
            if mids is not None and wsp is not None and all(isinstance(x, hip.SplitRows) for x in xs):
                nats = [clip_forward.native_of(plan.graph, ch.trie, li, li + 1) for ch in chunks]
                if all(n is not None for n in nats):       # fc2 + residual add + the next layer's LN1: one C call per slice
                    return [hip.clip_layer_tail(n.array, li, n.h, n.d, x, mid, next_ln) for n, x, mid in zip(nats, xs, mids)]
            if mids is None:
                return [lin(x, m.weight, m.bias, wsp=wsp) for x in xs]
            return [lin(x, m.weight, m.bias, residual=mid, wsp=wsp) for x, mid in zip(xs, mids)]

        with torch.no_grad():
            # the unedited leading layers: already launched by prepare (underneath the host's tokenization), else here
            states = []
            for ch in chunks:
                if ch.state is not None and ch.state[0] == first_edit:
                    states.append((ch.state[1], ch.state[2]))
                else:
                    states.append(clip_forward.run_prefix(plan.graph, ch.trie, first_edit))
                ch.state = None         # single use: the residual stream below belongs to the weights of this very call
            clip_forward.LAST_PATHS["forward_trie"] += 1
            clip_forward.run_layers_multi(plan.graph, [ch.trie for ch in chunks], states, first_edit, last, on_fc2,
                                          fc2_by_callback=order, callback_adds_residual=True, split_aware=True)
    else:
        def make_hook(i, layer):
            def hook(mod, inputs, output):
                x = inputs[0]
                solve(i, layer, gather_request_means(x, plan.ensure_batch()), gather_request_means(output, plan.ensure_batch()))
                if layer == last:
                    raise StopForward()
                return clip_forward.linear(x.reshape(-1, x.shape[-1]), mod.weight, mod.bias).reshape(*x.shape[:-1], -1) \
                    if x.is_cuda and x.dtype == torch.float32 else F.linear(x, mod.weight, mod.bias)
            return hook

        clip_forward.LAST_PATHS["forward_hf"] += 1
        for i, l in enumerate(plan.layers):
            handles.append(mods[l].register_forward_hook(make_hook(i, l)))
        try:
            with torch.no_grad(), hip_attention(te):
                try:
                    te(**plan.ensure_batch().inputs)
                except StopForward:
                    pass
        finally:
            for hd in handles:
                hd.remove()
    if len(edits) != L:
        raise RuntimeError(f"only {len(edits)} of {L} edited layers were reached by the forward pass "
                           f"(hparams.layers must be in forward order)")
    if dual and plan.factor_key is not None and _factor_cache_size() > 0:
        # this pass factored lam*C' itself: finish the explicit inverse factors it did not need (off the critical path,
        # on the side stream) so that check_info can hand the complete set to later edits
        missing = [i for i in range(L) if i not in plan.cov_factors.have_inverse]
        if missing:
            plan.side_stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(plan.side_stream):
                for i in missing:
                    hip.cov_inverse(plan.cov_factors, i, 1)
                plan.cov_factors.ready = torch.cuda.Event()
                plan.cov_factors.ready.record(plan.side_stream)
    if restore:
        with torch.no_grad():
            for l, w in weights.items():
                w.copy_(backups[l])
    if guard is not None:
        guard.flush()       # fingerprints of every weight a cache entry was made from in this pass (the edited fc2's as they are NOW)
    return edits


def solver_info(plan: EncoderEditPlan) -> int:
    """One host sync: 0, or 1 + the column of the first non-positive pivot any factorization of the last run met."""
    infos = [holder.info for holder in (plan.ws, plan.dual_ws, plan.cov_factors) if holder is not None]
    guard = plan.graph.guard if getattr(plan, "graph", None) is not None else None
    plan.stale_weights = False
    if guard is not None:
        if infos and infos[0].is_cuda and infos[0].device == guard.flag.device:
            infos = infos + [guard.flag]          # the stale-cache flag rides in the same read-back
        else:
            plan.stale_weights = bool(int(guard.flag.item()))
            guard = None
    if not infos:
        return 0
    if len(infos) == 1 or not infos[0].is_cuda:
        vals = [int(t.item()) for t in infos]
    else:       # all the flag words through ONE synchronisation: asynchronous copies into a pinned buffer, then one stream sync
        dev = infos[0].device
        host = torch.empty(len(infos), dtype=infos[0].dtype, pin_memory=True)
        for i, t in enumerate(infos):
            host[i:i + 1].copy_(t.reshape(-1)[:1], non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        vals = [int(v) for v in host.tolist()]
    if guard is not None:
        plan.stale_weights = bool(vals.pop())
    return next((v for v in vals if v != 0), 0)


def run_checked(plan: EncoderEditPlan, keep_factors: bool = False, restore: bool = False) -> List[LayerEdit]:
    """run_encoder_edit + check_info with the reference's fallback semantics: if a Cholesky factorization met a
    non-positive pivot (lam*C' + K K^T not positive definite, e.g. statistics that lost definiteness in fp32), the
    weights are put back and the whole pass is rerun with LU + partial pivoting — torch.linalg.solve's algorithm
    (reference emcid_main.py:1045), which returns numbers for any nonsingular system.  EMCID_LU_FALLBACK=0 raises instead."""
    edits = run_encoder_edit(plan, keep_factors=keep_factors, restore=restore)
    try:
        check_info(plan)
    except FloatingPointError:
        if os.environ.get("EMCID_LU_FALLBACK", "1") == "0" or plan.solver == "lu":
            raise
        plan.solver = "lu"
        plan.cov_factors = None
        clip_forward.LAST_PATHS["lu_fallbacks"] = clip_forward.LAST_PATHS.get("lu_fallbacks", 0) + 1
        logging.getLogger("emcid_amd").warning("a Cholesky factorization met a non-positive pivot: the pass is rerun with the pivoted-LU solver")
        edits = run_encoder_edit(plan, keep_factors=keep_factors, restore=restore)
        check_info(plan)
    return edits


def check_info(plan: EncoderEditPlan, restore_on_failure: bool = True):
    """One host sync at the very end: did any factorization meet a non-positive pivot?  If so the edited weights hold
    garbage: they are put back to the values they had before the run, then FloatingPointError is raised (callers that can
    retry — emcid_main — catch it and rerun with the pivoted-LU solver, the reference's own semantics)."""
    code = solver_info(plan)          # (a host synchronisation: nothing of this plan's run is still using its workspaces)
    _release_workspaces(plan)
    if getattr(plan, "stale_weights", False):
        # a weight was rewritten without its version counter moving: this pass ran on planes of the OLD bytes
        if plan.backups is not None:
            with torch.no_grad():
                for l, w0 in plan.backups.items():
                    get_parameter(plan.text_encoder, plan.weight_name(l)).copy_(w0)
        clip_forward.invalidate_weight_caches(plan.text_encoder)
        plan.factor_key, plan.graph = None, None
        raise clip_forward.StaleWeightCacheError(
            "an encoder weight was written in a way torch's version counter does not see (param.data.copy_/add_, a raw pointer) "
            "after the forward's caches were made from it; the edited weights have been put back and the caches dropped — call "
            "again (emcid_main's entry points do so by themselves), and bump the counter or call "
            "emcid_amd.invalidate_weight_caches(text_encoder) after such writes")
    if code != 0 and plan.solver == "lu":
        if restore_on_failure and plan.backups is not None:
            with torch.no_grad():
                for l, w0 in plan.backups.items():
                    get_parameter(plan.text_encoder, plan.weight_name(l)).copy_(w0)
        raise torch.linalg.LinAlgError(
            f"lam*C + K K^T is singular to working precision (zero pivot at column {code - 1} of the pivoted LU): "
            f"torch.linalg.solve (reference emcid_main.py:1045) raises for this system too; the edited weights have been restored")
    if code == 0:
        if plan.factor_key is not None and plan.cov_factors is not None and _factor_cache_size() > 0:
            with ENGINE_LOCK:
                plan.cov_factors.cached = True
                _FACTOR_CACHE[plan.factor_key] = (plan.cov_factors, [plan.covs[l] for l in plan.layers])
                while len(_FACTOR_CACHE) > _factor_cache_size():
                    _FACTOR_CACHE.popitem(last=False)
            plan.factor_key = None
        return
    if plan.factor_key is not None:
        plan.factor_key = None
        plan.cov_factors = None
    if restore_on_failure and plan.backups is not None:
        with torch.no_grad():
            for l, w0 in plan.backups.items():
                get_parameter(plan.text_encoder, plan.weight_name(l)).copy_(w0)
    raise FloatingPointError(
        f"lam*C + K K^T is not positive definite (non-positive pivot at column {code - 1}); the reference's LU "
        f"(torch.linalg.solve, emcid_main.py:1045) returns numbers for an indefinite system — the edited weights have been "
        f"restored; rerun with solver='lu' (emcid_main does so by itself) or check the statistics file / mom2_update_weight")
