"""Drop-in boundary of the closed-form mass-edit path on MI355X.

Mirrors the reference's entry points with the same names, argument meaning, side effects and returns
(reference: emcid/emcid_main.py):
    apply_emcid_to_text_encoder         :769-815      execute_emcid_text_encoder         :818-1082
    apply_emcid_to_sdxl_text_encoders   :38-106       execute_emcid_sd_xl_text_encoders  :1085-1425
    get_cov_text_encoder                :2239-2276    upd_matrix_match_shape             :2279-2298
    apply_emcid_to_cross_attn           :511-548      execute_emcid_cross_attn           :314-508
    get_cov_cross_attn                  :2203-2236    cal_insert_deltas                  :1969-2052
plus ``apply_emcid_to_model`` (the name BASELINE.json uses; dispatches on the hparams type).

What runs where: tokenizing, subject search, v*/C cache reads are host Python (once per call); everything
between "inputs are in HBM" and "fc2 weights are edited" is edit_engine.run_encoder_edit -> HIP kernels.
Kept reference behaviours: ``hparams`` is mutated in place by the mom2/edit weight overrides (:846-847);
``requests`` is never written (the reference deep-copies it for that, :850); ``execute_*`` leaves TE1 weights untouched (:1076-1078); the SDXL path
leaves TE2 edited and ``apply_*`` then adds the deltas again, so TE2 ends at W + 2*dW (:1410 vs :93-99) —
reproduced by default (``SDXL_TE2_DOUBLE_APPLY``).  Deliberate differences: ``COV_CACHE`` is keyed by the
statistics directory too (the reference's key ignores it and silently reuses a stale C);
per-request progress prints obey ``verbose``; a v* cache miss runs Stage 1 (compute_z.compute_z_text_encoder) when the
pipeline carries a UNet and a VAE (SDXL: compute_z.compute_z_sdxl_text_encoders, both vectors in one optimisation), and raises
otherwise.
"""
import functools
import logging
import os
import time
from copy import deepcopy
from pathlib import Path
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import clip_forward, edit_engine, hip, nethook
from .edit_engine import (ConceptShard, EncoderEditPlan, LayerEdit, check_info, phase, prepare_encoder_edit,
                          run_checked, run_encoder_edit)
from .emcid_hparams import EMCIDHyperParams, EMCIDXLHyperParams
from .globals import STATS_DIR, XL_STATS_DIR1, XL_STATS_DIR2
from .compute_ks import get_layers_input_output_at_words_cross_attn
from .layer_stats import get_all_cross_attn_kv_layer_names, layer_stats_cross_attn_kv, layer_stats_text_encoder

COV_CACHE: Dict[tuple, object] = {}                # key -> (d, d) fp32 on cpu, like the reference's (:36), or a _HostMoment that makes it on demand
_COV_DEVICE_CACHE: Dict[tuple, torch.Tensor] = {}  # (key, device) -> the same matrix resident in HBM
_VSTAR_CACHE: Dict[tuple, np.ndarray] = {}         # (path, mtime_ns, size) -> v_star
SDXL_TE2_DOUBLE_APPLY = True                       # reference quirk, see module docstring

Stage1Fn = Callable[..., torch.Tensor]


# ---- statistics ---------------------------------------------------------------------------------------

_RESOLVED_DIRS: Dict[Tuple[str, str], str] = {}     # (cwd, stat_dir as given) -> resolved path (Path.resolve is a chain of readlinks)


def _resolved(stat_dir) -> str:
    k = (os.getcwd(), str(stat_dir))
    r = _RESOLVED_DIRS.get(k)
    if r is None:
        if len(_RESOLVED_DIRS) > 256:
            _RESOLVED_DIRS.clear()
        r = _RESOLVED_DIRS[k] = str(Path(stat_dir).resolve())
    return r


def _cov_key(model, layer_name, stat_dir, mom2_n_samples, mom2_dtype):
    model_name = model.config._name_or_path.replace("/", "_")
    return (model_name, layer_name, _resolved(stat_dir), mom2_n_samples, mom2_dtype)


class _HostMoment:
    """COV_CACHE entry of a statistic that went from its file straight to the GPU (``_cov_from_file``): the host tensor the
    reference keeps there (mom2 / count, fp32) is fetched back from the device copy if anyone ever asks for it."""

    def __init__(self, on_device: torch.Tensor):
        self.on_device, self._c = on_device, None

    def tensor(self) -> torch.Tensor:
        if self._c is None:
            self._c = self.on_device.cpu()
        return self._c


def _host_cov(entry) -> torch.Tensor:
    return entry.tensor() if isinstance(entry, _HostMoment) else entry


def _cov_from_file(path, sample_size, device):
    """(C on ``device``, COV_CACHE entry) of a float32 second-moment file in the reference's npz format, or None (no such file,
    another layout or dtype, recorded sample_size differs: the general path then loads — or computes — it).  The file is read
    into ONE page-locked buffer (no zipfile, no CRC pass: runningstats.read_npz_stored), its mom2 member uploaded from there
    and divided by the count on the GPU — element-wise IEEE division by a device scalar, the same fp32 quotients as the
    host's ``mom2 / count`` (tests/test_e2e_gpu.py) — instead of: read, divide into a fresh 37.7 MB host tensor, pageable
    upload, each paying first-touch page faults (27 ms per layer in a cold process on the test box; 5 ms this way)."""
    from . import runningstats as rs
    if torch.device(device).type != "cuda" or not rs.global_load_cache_enabled or os.environ.get("EMCID_COV_FAST", "1") == "0":
        return None
    keep = {}

    def alloc(nbytes):
        keep["t"] = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        return keep["t"].numpy()

    try:
        dat = rs.read_npz_stored(path, alloc=alloc)
    except RuntimeError:            # no page-locked memory to be had: the general path reads into pageable memory
        return None
    if dat is None:
        return None
    try:
        dat = rs.unbox_numpy_null(dat)
        m, count = dat["mom2.mom2"], int(dat["mom2.count"])
        if sample_size is not None and dat.get("sample_size") != sample_size:
            return None
    except (KeyError, TypeError, ValueError):
        return None
    if m.dtype != np.float32 or m.ndim != 2 or m.shape[0] != m.shape[1] or count <= 0:
        return None
    # m is a view of the page-locked image: its bytes are uploaded as a slice of that very tensor (so the caching host allocator
    # knows the block is in use until the copy has run) and re-typed on the device, where the allocation is aligned
    image = keep["t"]
    off = m.__array_interface__["data"][0] - image.data_ptr()
    if off < 0 or off + m.nbytes > image.numel() or not m.flags.c_contiguous:
        return None
    dev = image[off:off + m.nbytes].to(device, non_blocking=True).view(torch.float32).reshape(m.shape)
    c = torch.div(dev, torch.full((), float(count), dtype=torch.float32, device=device))
    return c, _HostMoment(c)


def get_cov_text_encoder(model, tok, layer_name: str, mom2_dataset: str, mom2_n_samples: int, mom2_dtype: str,
                         inv: bool = False, force_recompute: bool = False, verbose: bool = True,
                         stat_dir=STATS_DIR) -> torch.Tensor:
    """Second moment C = mom2 / count of ``layer_name``'s input as fp32 on the model's device (loaded from the
    npz cache, else computed by Stage 0 over ./data/ccs_filtered.json)."""
    key = _cov_key(model, layer_name, stat_dir, mom2_n_samples, mom2_dtype)
    device = next(model.parameters()).device
    if verbose:
        print(f"Retrieving covariance statistics for {key[0]} @ {layer_name}.")
    dkey = (key, device)
    if key not in COV_CACHE or force_recompute:
        fast = None
        if not force_recompute and mom2_dtype == "float32":
            from .layer_stats import stats_filename
            fast = _cov_from_file(stats_filename(stat_dir, "text_encoder", mom2_dataset, layer_name, mom2_dtype, ["mom2"], 3 * 1024,
                                                 mom2_n_samples), mom2_n_samples, device)
        if fast is not None:
            _COV_DEVICE_CACHE[dkey], COV_CACHE[key] = fast
        else:
            stat = layer_stats_text_encoder(model, tok, layer_name, stat_dir, mom2_dataset, to_collect=["mom2"],
                                            sample_size=mom2_n_samples, precision=mom2_dtype,
                                            force_recompute=force_recompute)
            COV_CACHE[key] = stat.mom2.moment().float().to("cpu")
            _COV_DEVICE_CACHE.pop(dkey, None)
    if dkey not in _COV_DEVICE_CACHE:
        _COV_DEVICE_CACHE[dkey] = _host_cov(COV_CACHE[key]).to(device)
    c = _COV_DEVICE_CACHE[dkey]
    return torch.inverse(c) if inv else c


def clear_caches():
    COV_CACHE.clear()
    _COV_DEVICE_CACHE.clear()
    _VSTAR_CACHE.clear()


def upd_matrix_match_shape(matrix: torch.Tensor, shape: torch.Size) -> torch.Tensor:
    if matrix.shape == shape:
        return matrix
    if matrix.T.shape == shape:
        return matrix.T
    if matrix.dim() == 2 and len(shape) == 4:
        return matrix.reshape(shape[0], shape[1], *shape[2:])
    raise ValueError(f"Update matrix of shape {tuple(matrix.shape)} does not match the weight shape {tuple(shape)}")


# ---- v* cache -----------------------------------------------------------------------------------------

def vstar_cache_name(cache_name: Optional[str], request: Dict, hparams, idx: int, suffix: str = "") -> Optional[str]:
    """Cache path of one request's v* (reference :873-890 SD; :1157-1166 SDXL with suffix ``_2``)."""
    if cache_name is None:
        return None
    if "esd" in hparams.objective:
        return f"{cache_name}source_{request['source']}{suffix}.npz"
    if getattr(hparams, "sld_supervision", False):
        return f"{cache_name}source_{request['source_cat']}_{idx}{suffix}.npz"
    return f"{cache_name}source_{request['source']}_dest_{request['dest']}{suffix}.npz"


def vstar_cache_file(cache_name: Optional[str], request: Dict, hparams, idx: int, suffix: str = "") -> Optional[Path]:
    name = vstar_cache_name(cache_name, request, hparams, idx, suffix)
    return None if name is None else Path(name)


def _npz_single_array(blob: bytes, key: str) -> Optional[np.ndarray]:
    """The array of an npz whose FIRST member is ``{key}.npy``, stored uncompressed, little-endian f4/f8, C order — the
    file ``np.savez(f, v_star=...)`` writes (reference :951-968) — parsed straight from the bytes: the zip local header,
    then the npy header, then the data.  ``zipfile`` + ``np.load`` cost ~130 us per file, 1 000 files per mass edit.
    Returns None for anything else (the caller then uses ``np.load``)."""
    try:
        if blob[:4] != b"PK\x03\x04" or blob[8:10] != b"\x00\x00":          # local file header, method 0 = stored
            return None
        n_name, n_extra = int.from_bytes(blob[26:28], "little"), int.from_bytes(blob[28:30], "little")
        if blob[30:30 + n_name] != (key + ".npy").encode():
            return None
        o = 30 + n_name + n_extra
        if blob[o:o + 6] != b"\x93NUMPY":
            return None
        major = blob[o + 6]
        if major == 1:
            hlen, o = int.from_bytes(blob[o + 8:o + 10], "little"), o + 10
        elif major in (2, 3):
            hlen, o = int.from_bytes(blob[o + 8:o + 12], "little"), o + 12
        else:
            return None
        header = blob[o:o + hlen].decode("latin1")
        o += hlen
        import ast
        meta = ast.literal_eval(header)
        descr, shape = meta["descr"], tuple(meta["shape"])
        if meta["fortran_order"] and len(shape) > 1 or descr not in ("<f4", "<f8"):
            return None
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        dt = np.dtype(descr)
        if o + n * dt.itemsize > len(blob):
            return None
        return np.frombuffer(blob, dtype=dt, count=n, offset=o).reshape(shape).copy()
    except Exception:
        return None


_VSTAR_CACHE_MAX = 4096     # entries of the per-file memo below (the numpy path only); oldest dropped first


def _read_vstar(path: str) -> np.ndarray:
    """``np.load(path)["v_star"]`` for ONE file — the path of every file the native batch reader does not serve (other npz
    layouts, compressed members) and of processes without libemcid_host.so; memoised on (path, mtime, size), bounded."""
    st = os.stat(path)
    key = (path, st.st_mtime_ns, st.st_size)
    v = _VSTAR_CACHE.get(key)
    if v is None:
        with open(path, "rb") as f:
            blob = f.read()
        v = _npz_single_array(blob, "v_star")
        if v is None:
            with np.load(path) as z:
                v = np.asarray(z["v_star"])
        while len(_VSTAR_CACHE) >= _VSTAR_CACHE_MAX:
            _VSTAR_CACHE.pop(next(iter(_VSTAR_CACHE)))
        _VSTAR_CACHE[key] = v
    return v


def _read_threads() -> int:
    n = os.environ.get("EMCID_READ_THREADS", "")
    if n:
        return max(1, int(n))
    from . import effective_cpu_count
    return max(1, min(8, effective_cpu_count() // 2))


def _native_vstar_rows(names: Sequence[Optional[str]], width: int, pin: bool):
    """All cache files in one native call (csrc/host_io.cpp: a few threads, open/read/parse straight into the row buffer —
    page-locked when the rows go to a GPU next, so the upload needs no staging copy).  Returns (rows (N, width) fp32 tensor,
    status uint8 array: 0 = row read, 1 = no such file, 2 = a file for numpy), or None without the host library."""
    from . import host_text
    if os.environ.get("EMCID_NATIVE_VSTAR", "1") == "0" or not host_text.available() or any(n is None for n in names):
        return None
    lib = host_text.load()
    n = len(names)
    blob, off = host_text.pack_strings(names)
    rows = torch.empty((n, width), dtype=torch.float32, pin_memory=bool(pin))
    status = np.empty(n, dtype=np.uint8)
    rc = lib.emcid_read_npz_rows_f32(blob, off.ctypes.data, n, b"v_star", width, rows.data_ptr(), width,
                                     status.ctypes.data, _read_threads())
    if rc < 0:
        return None
    return rows, status


def load_v_stars(requests: Sequence[Dict], hparams, cache_name: Optional[str], suffix: str = "",
                 stage1: Optional[Stage1Fn] = None, width: Optional[int] = None, pin: bool = False, native=None) -> torch.Tensor:
    """(N, hidden) fp32 on the host: one row per request, the transpose of the reference's ``zs`` (:977) — (N k, hidden), row
    ``rq * k + num``, for ``use_new_compute_z`` files of k = num_edit_tokens rows each (:972-975).  ``width``: the
    encoder's hidden size when the caller knows it (then every cache file is read by the native batch reader; files it does
    not serve, and every miss, take the per-file path below, which is the reference's: np.load, recompute on an unreadable
    file :903-904, Stage 1 on a miss :905-969).  ``native``: (rows, status) of a native batch read of these very names that has
    already been made (the early reader's), so that a miss does not read the hits a second time."""
    names = [vstar_cache_name(cache_name, request, hparams, idx, suffix) for idx, request in enumerate(requests)]
    if native is None:
        native = _native_vstar_rows(names, int(width), pin) if (width and cache_name is not None and len(names)) else None
    new_z = bool(getattr(hparams, "use_new_compute_z", False))
    k_tok = int(getattr(hparams, "num_edit_tokens", 1) or 1)
    if native is not None and not native[1].any() and not (new_z and k_tok > 1):
        # every file was read natively: (width,) or (1, width) rows, which is all a one-token edit can hold — a k-token edit's
        # files are (k, width) and take the per-file path below with its shape check (the native reader does not serve them)
        return native[0]
    rows: List[Optional[np.ndarray]] = [None] * len(requests)
    missing: List[int] = []
    for idx, request in enumerate(requests):
        f = names[idx]
        if native is not None and native[1][idx] == 0:
            rows[idx] = native[0][idx].numpy()
            continue
        v = None
        if f is not None and not (native is not None and native[1][idx] == 1):
            try:
                v = _read_vstar(f)
            except FileNotFoundError:
                v = None
            except Exception as e:  # unreadable cache -> recompute, as the reference (:903-904)
                print(f"Error reading cache file due to {e}. Recomputing...")
                v = None
        if v is None:
            if stage1 is None:
                raise NotImplementedError(
                    f"no cached v* for request {idx} ([{request['source']}] -> [{request['dest']}]) at {f}: "
                    f"pass cache_name pointing at v_star npz files (reference emcid_main.py:873-890) or a stage1= "
                    f"callable (emcid_amd.compute_z.compute_z_text_encoder is the reference's Stage 1)")
            missing.append(idx)
            continue
        rows[idx] = v
    if missing:
        # Stage 1 for every miss, in request order like the reference's loop (:871-969) — through the callable's ``batch``
        # attribute when it has one (compute_z.stage1_for: several concepts per Adam step), else one request at a time
        with torch.enable_grad():      # Stage 1 is an optimisation through the UNet, whatever mode the caller is in
            if hasattr(stage1, "batch") and len(missing) > 1:
                found = stage1.batch([requests[i] for i in missing], suffix)
            else:
                found = [stage1(requests[i], suffix) for i in missing]
        for idx, v in zip(missing, found):
            v = v.detach().float().cpu().numpy()
            if names[idx] is not None:
                Path(names[idx]).parent.mkdir(exist_ok=True, parents=True)
                np.savez(names[idx], v_star=v)
            rows[idx] = v
    new_z = bool(getattr(hparams, "use_new_compute_z", False))
    for idx, v in enumerate(rows):
        if v.dtype != np.float32:
            v = v.astype(np.float32)
        if new_z:         # files of shape (num_edit_tokens, hidden) (:946-957); zs = "rq num c_i -> c_i (rq num)" of their stack (:972-975)
            if v.ndim == 1:
                v = v[None, :]
            if v.ndim != 2 or v.shape[0] != int(hparams.num_edit_tokens):
                raise ValueError(f"v* of request {idx} has shape {tuple(v.shape)}; use_new_compute_z with num_edit_tokens = "
                                 f"{hparams.num_edit_tokens} expects ({hparams.num_edit_tokens}, hidden)")
        elif v.ndim == 2:
            if v.shape[0] != 1:
                raise ValueError(f"v* of request {idx} has {v.shape[0]} rows: a multi-token v* needs hparams.use_new_compute_z "
                                 f"(reference emcid_main.py:898-900, :972-977)")
            v = v[0]
        rows[idx] = v
    out = np.stack(rows, axis=0)
    return torch.from_numpy(out.reshape(-1, out.shape[-1]) if new_z else out)


# ---- plans --------------------------------------------------------------------------------------------

def _shard_from_env(shard: Optional[ConceptShard]) -> ConceptShard:
    if shard is not None:
        return shard
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        force = os.environ.get("EMCID_FORCE_COLLECTIVES", "0") == "1"
        if dist.get_world_size() > 1 or force:
            return ConceptShard(dist.get_rank(), dist.get_world_size(), None, force_collectives=force)
    return ConceptShard()


_DIR_LISTINGS: Dict[str, Tuple[int, frozenset]] = {}      # directory -> (its st_mtime_ns, its entries): a listing is re-read only
                                                           # when the directory changed (a file added or removed bumps mtime)


def _dir_entries(d: str) -> Optional[frozenset]:
    try:
        m = os.stat(d or ".").st_mtime_ns
        hit = _DIR_LISTINGS.get(d)
        if hit is None or hit[0] != m:
            hit = _DIR_LISTINGS[d] = (m, frozenset(os.listdir(d or ".")))
            if len(_DIR_LISTINGS) > 64:
                _DIR_LISTINGS.pop(next(iter(_DIR_LISTINGS)))
        return hit[1]
    except OSError:
        return None


def _any_vstar_missing(requests: Sequence[Dict], hparams, cache_name: Optional[str], suffix: str) -> bool:
    """True when some request has no v* file: one directory listing per cache directory (re-read only when the directory's
    mtime moved) instead of a stat per request; the per-file validation by mtime/size happens later, underneath the GPU's
    forward.  A stale answer is harmless either way: "missing" takes the eager path, which opens the files; "present" takes
    the lazy one, which handles a file that is gone exactly like the eager one, just later."""
    if cache_name is None:
        return True
    pre = str(cache_name)
    try:
        if "esd" in hparams.objective:
            tails = [f"source_{r['source']}{suffix}.npz" for r in requests]
        elif getattr(hparams, "sld_supervision", False):
            tails = [f"source_{r['source_cat']}_{i}{suffix}.npz" for i, r in enumerate(requests)]
        else:
            tails = [f"source_{r['source']}_dest_{r['dest']}{suffix}.npz" for r in requests]
    except KeyError:
        return True                                        # the eager path raises the reference's KeyError
    if any("/" in t for t in tails):                       # a source with a path separator: per-file check
        return not all(os.path.exists(pre + t) for t in tails)
    pre_dir, pre_base = os.path.split(pre)
    names = _dir_entries(pre_dir)
    if names is None:
        return True
    return not names.issuperset(pre_base + t for t in tails) if pre_base else not names.issuperset(tails)


_VSTAR_READER = None          # (pid, executor): one helper thread per process, created at first use


class _EarlyVstars:
    """The v* rows of a request list, read by the native batch reader ON A HELPER THREAD from the moment prepare has launched the
    unedited leading layers: the reader is one ctypes call that does not hold the interpreter lock, so it runs beside the host's
    remaining preparation and is (nearly) done when the first solve asks — with the GPU twice as fast as in round 3 the read had
    moved onto the critical path (profiles/r04_g_call_events.txt: the host reached the first solve 0.2 ms before the device).
    Anything the native reader does not serve (a miss, a file for numpy, k-token files) falls back to load_v_stars at result().
    The rows are a SNAPSHOT of the files as they are when ``prepare`` runs: a plan prepared early and run after its cache files were
    rewritten edits towards the old targets (the lazy reader, EMCID_EARLY_VSTAR=0, reads at the first solve)."""

    def __init__(self, args, kwargs, future, rows, keep):
        self.args, self.kwargs, self.future, self.rows, self.keep = args, kwargs, future, rows, keep

    @classmethod
    def start(cls, requests, hparams, cache_name, suffix, stage1, width=None, pin=False):
        from . import host_text
        if (not width or cache_name is None or not len(requests) or os.environ.get("EMCID_NATIVE_VSTAR", "1") == "0"
                or os.environ.get("EMCID_EARLY_VSTAR", "1") == "0" or not host_text.available()):
            return None
        if bool(getattr(hparams, "use_new_compute_z", False)) and int(getattr(hparams, "num_edit_tokens", 1) or 1) > 1:
            return None
        names = [vstar_cache_name(cache_name, request, hparams, idx, suffix) for idx, request in enumerate(requests)]
        if any(n is None for n in names):
            return None
        global _VSTAR_READER
        if _VSTAR_READER is None or _VSTAR_READER[0] != os.getpid():      # (a forked child does not inherit the worker thread)
            from concurrent.futures import ThreadPoolExecutor
            _VSTAR_READER = (os.getpid(), ThreadPoolExecutor(max_workers=1, thread_name_prefix="emcid-vstar"))
        lib = host_text.load()
        n = len(names)
        blob, off = host_text.pack_strings(names)
        rows = torch.empty((n, int(width)), dtype=torch.float32, pin_memory=bool(pin))
        status = np.empty(n, dtype=np.uint8)
        threads = _read_threads()

        times = [time.perf_counter(), 0.0, 0.0]      # submitted, started, finished (edit_engine.TIMING: "vstar reader ...")

        def read(blob=blob, off=off, rows=rows, status=status, times=times):
            # (the buffers belong to this task, not to the object that waits for it: a plan that is dropped unrun must not free
            #  memory the reader is still writing)
            times[1] = time.perf_counter()
            rc = lib.emcid_read_npz_rows_f32(blob, off.ctypes.data, n, b"v_star", int(width), rows.data_ptr(), int(width),
                                             status.ctypes.data, threads)
            times[2] = time.perf_counter()
            return rc

        fut = _VSTAR_READER[1].submit(read)
        self = cls((requests, hparams, cache_name, suffix, stage1), dict(width=width, pin=pin), fut, rows, (blob, off, status))
        self.times = times
        return self

    def wait(self):
        try:
            rc = self.future.result()
        except Exception as e:       # the per-file path below serves the call; say why the batch read did not
            logging.getLogger("emcid_amd").warning("native v* batch read failed (%r): reading the cache files one by one", e)
            return -1
        t = getattr(self, "times", None)
        if t is not None and t[2] > 0.0:       # where a slow join comes from: the worker thread starting late, or the reads themselves
            edit_engine.TIMING["vstar reader queued"] = edit_engine.TIMING.get("vstar reader queued", 0.0) + (t[1] - t[0])
            edit_engine.TIMING["vstar reader reading"] = edit_engine.TIMING.get("vstar reader reading", 0.0) + (t[2] - t[1])
            self.times = None
        return rc

    def result(self):
        rc = self.wait()
        if rc is not None and rc >= 0 and not self.keep[2].any():
            return self.rows
        # a miss or a file for numpy: the rows the batch read did serve are handed on (no second read of 999 hits for one miss)
        native = (self.rows, self.keep[2]) if rc is not None and rc >= 0 else None
        return load_v_stars(*self.args, **self.kwargs, native=native)


class _LazyVstars:
    """The v* rows of a request list, read when first asked for.  prepare hands this to the engine, which asks at the first
    edited layer's solve — by then the encoder forward up to that layer is queued on the GPU, so the file-system calls of
    the cache reads (and Stage 1 on a miss) cost no wall-clock of their own."""

    def __init__(self, *args, **kwargs):
        self.args, self.kwargs = args, kwargs

    def result(self):
        return load_v_stars(*self.args, **self.kwargs)


def prepare_text_encoder_edit(text_encoder, tokenizer, requests, hparams, layers, lam, stat_dir, cache_name,
                              suffix="", verbose=True, shard=None, stage1=None) -> EncoderEditPlan:
    """Host side of one encoder's edit: v* rows, C per layer (HBM-resident), tokenized prompts + lookup."""
    w = None
    for layer in layers:   # resolve every edited weight now: LookupError before any GPU work, like the reference (:858-863)
        w = nethook.get_parameter(text_encoder, f"{hparams.rewrite_module_tmp.format(layer)}.weight")
    # v* rows have the encoder's hidden size = the rows of the edited projection (h, d); known here, it lets the cache files be
    # read natively, in one call, into page-locked rows (load_v_stars)
    how = dict(width=int(w.shape[0]) if w is not None and w.dim() == 2 else None, pin=bool(w is not None and w.is_cuda))

    def targets():
        # the files start coming in on the helper thread at once; whether one is missing is looked up meanwhile
        early = _EarlyVstars.start(requests, hparams, cache_name, suffix, stage1, **how)
        # a cache miss is handled FIRST, as the reference does (:873-969 come before the layer loop's covariance reads): Stage 1
        # runs (or the miss is reported) before statistics are read or computed
        if _any_vstar_missing(requests, hparams, cache_name, suffix):
            if early is not None:
                return early.result()   # waits for the reader, hands the rows it did read to load_v_stars (Stage 1 for the misses)
            return load_v_stars(requests, hparams, cache_name, suffix, stage1, **how)
        return early if early is not None else _LazyVstars(requests, hparams, cache_name, suffix, stage1, **how)

    def statistics():
        return {layer: get_cov_text_encoder(text_encoder, tokenizer, hparams.rewrite_module_tmp.format(layer),
                                            hparams.mom2_dataset, hparams.mom2_n_samples, hparams.mom2_dtype,
                                            stat_dir=stat_dir, verbose=verbose)
                for layer in layers}

    # both run inside prepare_encoder_edit right AFTER it has launched the unedited leading layers (they need nothing but the
    # prompts), in this order
    return prepare_encoder_edit(text_encoder, tokenizer, requests, layers, hparams.rewrite_module_tmp, lam,
                                hparams.edit_weight, targets, statistics, _shard_from_env(shard),
                                layer_module_tmp=getattr(hparams, "layer_module_tmp", None),
                                num_edit_tokens=int(getattr(hparams, "num_edit_tokens", 1)))


def _default_stage1(pipe, hparams, stage1):
    """Stage 1 on a v* cache miss, like the reference (:905-969): when the caller gave no ``stage1=`` and the pipeline
    carries a UNet and a VAE, the missing v* is optimised by compute_z.compute_z_text_encoder at z_layer =
    hparams.layers[-1] (:868) and written to the cache — by compute_z_text_encoder_global under ``sld_supervision`` (:911-918:
    a global concept at "[CLS]" / "[EOS]" under the safe-latent-diffusion supervision), by compute_z_text_encoder_v1 when
    ``txt_img_align_scale_factor != 0`` (:919-926: CLIP's text tower with projection + the image-alignment term; its towers come
    from the hub like the reference's — ``stage1=compute_z.stage1_for(pipe, hparams, layer, clip_towers=...)`` hands over local
    ones), by compute_z_text_encoder_v2, (num_edit_tokens, hidden) per concept, under ``use_new_compute_z`` (:927-936)."""
    if stage1 is not None or getattr(pipe, "unet", None) is None or getattr(pipe, "vae", None) is None:
        return stage1
    from .compute_z import stage1_for
    return stage1_for(pipe, hparams, hparams.layers[-1])


def _default_stage1_sdxl(pipe, hparams, stage1):
    """Stage 1 of the SDXL pair on a v* cache miss, like the reference (:1157-1230): without a caller-supplied ``stage1=`` and
    with a UNet and a VAE in the pipeline, compute_z.compute_z_sdxl_text_encoders optimises (v*, v*_2) together at the last
    edited layer of each encoder; each encoder's loader writes its own cache file."""
    if stage1 is not None or getattr(pipe, "unet", None) is None or getattr(pipe, "vae", None) is None:
        return stage1
    from .compute_z import stage1_for_sdxl
    return stage1_for_sdxl(pipe, hparams)


def _deltas_to_host(edits: List[LayerEdit]) -> Dict[str, Tuple[torch.Tensor, torch.Tensor]]:
    """Reference return format: {weight_name: (adj_k (d, N) f64 cpu, resid (h, N) f64 cpu)} (:1062-1065)."""
    return {e.weight_name: (e.Xt.t().contiguous().cpu(), e.Rt.t().contiguous().cpu()) for e in edits}


def _announce(requests, verbose):
    if verbose:
        for request in requests:
            print(f"EMCID request sample: [{request['source']}] -> [{request['dest']}]")


def _any_rank(flag: bool, device) -> bool:
    """True on every rank of the default group if ``flag`` is set on any of them (one MAX all-reduce of a word)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32,
                     device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.item()))


def _retry_if_stale(fn):
    """The forward's weight-derived caches carry a content guard (clip_forward.WeightGuard): when an edit finds that a weight was
    rewritten behind them (``param.data.copy_(...)`` between two calls), the engine puts the edited weights back, drops the caches
    and raises ``StaleWeightCacheError`` — the call is redone once, from the live weights.  A multi-rank job raises instead: one
    rank redoing its call alone would leave the others inside their collectives."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        try:
            return fn(*args, **kwargs)
        except clip_forward.StaleWeightCacheError as e:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                raise
            logging.getLogger("emcid_amd").warning("%s: %s; redoing the call from the live weights", fn.__name__, e)
            clip_forward.invalidate_weight_caches(None)
            clip_forward.LAST_PATHS["stale_cache_retries"] = clip_forward.LAST_PATHS.get("stale_cache_retries", 0) + 1
            return fn(*args, **kwargs)
    return wrapper


# ---- SD ------------------------------------------------------------------------------------------------

@_retry_if_stale
def execute_emcid_text_encoder(pipe, requests: List[Dict], hparams: EMCIDHyperParams, cache_name: Optional[str] = None,
                               mom2_weight: Optional[int] = None, edit_weight: Optional[float] = None,
                               verbose: bool = True, stat_dir=STATS_DIR, shard=None, stage1=None
                               ) -> Dict[str, Tuple[torch.Tensor, torch.Tensor]]:
    """Computes the per-layer factors; the model is unchanged on return (invariant of the reference)."""
    hparams.mom2_update_weight = mom2_weight if mom2_weight is not None else hparams.mom2_update_weight
    hparams.edit_weight = edit_weight if edit_weight is not None else hparams.edit_weight
    _announce(requests, verbose)
    plan = prepare_text_encoder_edit(pipe.text_encoder, pipe.tokenizer, requests, hparams, hparams.layers,
                                     hparams.mom2_update_weight, stat_dir, cache_name, "", verbose, shard,
                                     _default_stage1(pipe, hparams, stage1))
    edits = run_checked(plan, keep_factors=True, restore=True)
    if verbose:
        print(f"Deltas successfully computed for {[e.weight_name for e in edits]}")
    return _deltas_to_host(edits)


@_retry_if_stale
def apply_emcid_to_text_encoder(pipe, requests: List[Dict], hparams: EMCIDHyperParams, device: str,
                                mom2_weight: Optional[int] = None, edit_weight: Optional[float] = None,
                                return_orig_text_encoder=False, cache_name: Optional[str] = None,
                                stats_dir=STATS_DIR, verbose: bool = True, shard=None, stage1=None):
    """Returns (pipe with the edited text encoder, the original text encoder or None)."""
    origin_text_encoder = deepcopy(pipe.text_encoder) if return_orig_text_encoder else None
    hparams.mom2_update_weight = mom2_weight if mom2_weight is not None else hparams.mom2_update_weight
    hparams.edit_weight = edit_weight if edit_weight is not None else hparams.edit_weight
    _announce(requests, verbose)
    plan = prepare_text_encoder_edit(pipe.text_encoder, pipe.tokenizer, requests, hparams, hparams.layers,
                                     hparams.mom2_update_weight, stats_dir, cache_name, "", verbose, shard,
                                     _default_stage1(pipe, hparams, stage1))
    # The engine leaves each fc2 at W0 + float(U): the value the reference reaches by restoring W0 (:1076-1078)
    # and adding float(adj_k @ resid^T) again (:802-809).
    with phase("run + final sync"):
        edits = run_checked(plan, keep_factors=False, restore=False)
    if verbose:
        print(f"New weights successfully inserted into {[e.weight_name for e in edits]}")
    return pipe, origin_text_encoder


@_retry_if_stale
def cal_insert_deltas(pipe, weights: Dict[str, torch.Tensor], hparams: EMCIDHyperParams, requests: List[Dict],
                      zs: torch.Tensor, verbose: bool = True, stat_dir=STATS_DIR, shard=None):
    """The Stage-2 layer loop for targets the caller already has (reference: :1969-2052, used by the debias driver):
    ``zs`` is (hidden, N), one column per request.  Returns {weight_name: (adj_k, resid)} and, like the reference, LEAVES
    the edited weights in the model (its callers restore from their own copies); ``weights`` must be the live
    ``rewrite_module_tmp`` parameters of ``pipe.text_encoder`` (it is what the reference indexes, :2031-2039)."""
    for layer in hparams.layers:
        name = f"{hparams.rewrite_module_tmp.format(layer)}.weight"
        if weights[name] is not nethook.get_parameter(pipe.text_encoder, name):
            raise ValueError(f"weights[{name!r}] is not the text encoder's live parameter")
    zs_t = zs.detach().t().contiguous().float().cpu()
    covs = {layer: get_cov_text_encoder(pipe.text_encoder, pipe.tokenizer, hparams.rewrite_module_tmp.format(layer),
                                        hparams.mom2_dataset, hparams.mom2_n_samples, hparams.mom2_dtype,
                                        stat_dir=stat_dir, verbose=verbose) for layer in hparams.layers}
    plan = prepare_encoder_edit(pipe.text_encoder, pipe.tokenizer, requests, hparams.layers, hparams.rewrite_module_tmp,
                                hparams.mom2_update_weight, hparams.edit_weight, zs_t, covs, _shard_from_env(shard),
                                layer_module_tmp=getattr(hparams, "layer_module_tmp", None))
    edits = run_checked(plan, keep_factors=True, restore=False)
    return _deltas_to_host(edits)


# ---- cross-attention K/V of the UNet (reference: :314-548) -------------------------------------------------------

def get_cov_cross_attn(pipe, layer_name: str, mom2_dataset: str, sample_size: int, mom2_dtype: str, inv: bool = False,
                       force_recompute: bool = False, verbose: bool = False, stats_dir=STATS_DIR) -> torch.Tensor:
    """Second moment of a cross-attention projection's input (the text embedding), fp32 on the UNet's device
    (reference: :2203-2236; its cache key ignores the statistics directory, this one includes it)."""
    model_name = pipe.unet.config._name_or_path.replace("/", "_")
    key = (model_name, layer_name, str(Path(stats_dir).resolve()), sample_size, mom2_dtype)
    device = next(pipe.unet.parameters()).device
    if verbose:
        print(f"Retrieving covariance statistics for {model_name} @ {layer_name}.")
    if key not in COV_CACHE or force_recompute:
        stat = layer_stats_cross_attn_kv(pipe, layer_name, stats_dir, mom2_dataset, to_collect=["mom2"],
                                         sample_size=sample_size, precision=mom2_dtype, force_recompute=force_recompute)
        COV_CACHE[key] = stat.mom2.moment().float().to("cpu")
        _COV_DEVICE_CACHE.pop((key, device), None)
    dkey = (key, device)
    if dkey not in _COV_DEVICE_CACHE:
        _COV_DEVICE_CACHE[dkey] = COV_CACHE[key].to(device)
    c = _COV_DEVICE_CACHE[dkey]
    return torch.inverse(c) if inv else c


def load_v_stars_cross_attn(requests: Sequence[Dict], cache_name: Optional[str], layer_names: Sequence[str],
                            stage1: Optional[Callable[[Dict], Dict[str, torch.Tensor]]] = None) -> Dict[str, torch.Tensor]:
    """{layer_name: (N, out) fp32 on the host}.  Cache: one npz per request, ``source_{source}.npz``, whose entries are
    pickled ``{"v_star": array}`` per layer name (reference: :373-391 read, :411-420 write)."""
    rows = {n: [] for n in layer_names}
    for idx, request in enumerate(requests):
        f = Path(cache_name + f"source_{request['source']}.npz") if cache_name is not None else None
        got = None
        if f is not None and f.exists():
            try:
                data = np.load(f, allow_pickle=True)
                got = {n: np.asarray(data[n].item()["v_star"], dtype=np.float32) for n in layer_names}
            except Exception as e:   # unreadable cache -> recompute, as the reference (:392-393)
                print(f"Error reading cache file due to {e}. Recomputing...")
        if got is None:
            if stage1 is None:
                raise NotImplementedError(
                    f"no cached cross-attention v* for request {idx} ([{request['source']}]) at {f} and no way to compute it: "
                    f"Stage 1 (compute_z_unet_x_kv) needs a pipeline with a UNet and a VAE — pass cache_name pointing at "
                    f"the reference's npz files or a stage1= callable")
            got = {n: v.detach().float().cpu().numpy() for n, v in stage1(request).items()}
            if f is not None:
                f.parent.mkdir(exist_ok=True, parents=True)
                np.savez(f, **{n: {"v_star": got[n]} for n in layer_names})
        for n in layer_names:
            rows[n].append(got[n])
    return {n: torch.from_numpy(np.stack(v, axis=0)) for n, v in rows.items()}


def _edit_cross_attn(pipe, requests, hparams, cache_name, stats_dir, keep_factors, restore, verbose, stage1):
    """Closed form for every cross-attention K/V projection.  Same keys for all of them (the text embedding at the last
    subject token), own targets, own statistics, residual NOT split over layers (:473).  Each projection is one
    ``emcid_edit_layer_f64`` call: d = text hidden size (768), N concepts, h = the block's channel count.

    One reference behaviour is NOT reproduced: ``apply_emcid_to_cross_attn`` forms ``adj_k @ resid^T`` (hidden x out) and
    relies on ``upd_matrix_match_shape`` to transpose it (:540-543); for a SQUARE projection the shape test passes
    untransposed and the reference adds U^T.  No Stable Diffusion UNet has channels == text hidden size (768 vs
    320/640/1280; 2048 vs 640/1280 for SDXL); here such a projection gets the intended U."""
    names = get_all_cross_attn_kv_layer_names(pipe)
    weights = {n: nethook.get_parameter(pipe.unet, f"{n}.weight") for n in names}
    device = next(pipe.unet.parameters()).device
    for n, w in weights.items():
        if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()):
            raise hip.EmcidHipError(f"{n}.weight must be a contiguous fp32 tensor in HBM (got {w.dtype} on {w.device})")
    if stage1 is None and getattr(pipe, "unet", None) is not None and getattr(pipe, "vae", None) is not None:
        # a v* miss runs Stage 1 of this sibling on the caller's UNet / VAE, like the reference (:398)
        from .compute_z import compute_z_unet_x_kv
        device = next(pipe.unet.parameters()).device
        stage1 = lambda request: compute_z_unet_x_kv(pipe, request, hparams, device)
    zs = load_v_stars_cross_attn(requests, cache_name, names, stage1)
    covs = {n: get_cov_cross_attn(pipe, n, hparams.mom2_dataset, hparams.mom2_n_samples, hparams.mom2_dtype,
                                  verbose=verbose, stats_dir=stats_dir) for n in names}
    ks, cur = get_layers_input_output_at_words_cross_attn(pipe, requests, names,
                                                          layer_module_tmp=getattr(hparams, "layer_module_tmp", None))
    out, infos = {}, []
    # The keys are the same for every projection, and so is the system matrix lam*C' + K K^T whenever two projections
    # share their statistics (they all see the same text embeddings: the reference's files differ in name only).  Such
    # projections share ONE assembly + factorization + solve (adj_k); each then costs its residual and one dW GEMM.
    groups: List[Tuple[torch.Tensor, List[str]]] = []
    for n in names:
        for rep, members in groups:
            if rep is covs[n] or (rep.shape == covs[n].shape and torch.equal(rep, covs[n])):
                members.append(n)
                break
        else:
            groups.append((covs[n], [n]))
    s_ = (hparams.edit_weight / 0.5) ** 0.5
    for cov, members in groups:
        lead = members[0]
        w = weights[lead]
        K = ks[lead].contiguous()
        w0 = w.detach().clone()
        if verbose:
            print(f"Writing {K.shape[0]} key/value pair(s) into layer {lead}")
        res = hip.edit_layer(K, cur[lead].contiguous(), zs[lead].to(device).contiguous(), cov, hparams.mom2_update_weight,
                             hparams.edit_weight, 1, W0=w0, W=w.data, want_factors=True, want_dw=False)
        infos.append(res["ws"].info)
        Xt = res["Xt"]                                             # (N, hidden) f64 = adj_k^T, shared by the group
        adj_k_host = Xt.t().contiguous().cpu() if keep_factors else None
        if keep_factors:
            out[f"{lead}.weight"] = (adj_k_host, res["Rt"].t().contiguous().cpu())
        if restore:
            w.data.copy_(w0)
        for n in members[1:]:
            w = weights[n]
            if verbose:
                print(f"Writing {K.shape[0]} key/value pair(s) into layer {n}")
            # resid^T = double(zs - Zc) * sqrt(e_w / 0.5)  (fp32 difference first, as :440 and :466)
            Rt = ((zs[n].to(device) - cur[n]).double() * s_).contiguous()
            if keep_factors:
                out[f"{n}.weight"] = (adj_k_host, Rt.t().contiguous().cpu())
            if not restore:
                hip.delta_w_(Rt, Xt, w.detach().clone(), w.data)
    out = {f"{n}.weight": out[f"{n}.weight"] for n in names if f"{n}.weight" in out}     # reference key order
    code = int(torch.stack(infos).max().item()) if infos else 0
    if code != 0:
        raise FloatingPointError(f"lam*C + K K^T is not positive definite (non-positive pivot at column {code - 1})")
    return names, out


def execute_emcid_cross_attn(pipe, requests: List[Dict], hparams: EMCIDHyperParams, cache_name: Optional[str] = None,
                             mom2_weight: Optional[int] = None, edit_weight: Optional[float] = None, verbose: bool = True,
                             stats_dir=STATS_DIR, stage1=None) -> Dict[str, Tuple[torch.Tensor, torch.Tensor]]:
    """{weight_name: (adj_k (hidden, N) f64 cpu, resid (out, N) f64 cpu)}; the UNet is unchanged on return (:314-508)."""
    hparams.mom2_update_weight = mom2_weight if mom2_weight is not None else hparams.mom2_update_weight
    hparams.edit_weight = edit_weight if edit_weight is not None else hparams.edit_weight
    requests = deepcopy(requests)
    if verbose:
        for request in requests:
            print(f"EMCID request sample: [{request['source']}] -> [{request['dest']}]" if "dest" in request
                  else f"EMCID request sample: erasing [{request['source']}]")
    names, deltas = _edit_cross_attn(pipe, requests, hparams, cache_name, stats_dir, True, True, verbose, stage1)
    if verbose:
        print(f"Deltas successfully computed for {[f'{n}.weight' for n in names]}")
    return deltas


def apply_emcid_to_cross_attn(pipe, requests: List[Dict], hparams: EMCIDHyperParams, device: str,
                              mom2_weight: Optional[int] = None, edit_weight: Optional[float] = None,
                              return_orig_text_model=False, cache_name: Optional[str] = None, stats_dir=STATS_DIR,
                              verbose: bool = True, stage1=None):
    """Returns (pipe with the edited UNet projections, the original UNet or None) (:511-548)."""
    orig_unet = deepcopy(pipe.unet) if return_orig_text_model else None
    hparams.mom2_update_weight = mom2_weight if mom2_weight is not None else hparams.mom2_update_weight
    hparams.edit_weight = edit_weight if edit_weight is not None else hparams.edit_weight
    requests = deepcopy(requests)
    # each projection is left at W0 + float(U): what the reference reaches by restoring W0 and adding
    # float(adj_k @ resid^T) (:538-545)
    names, _ = _edit_cross_attn(pipe, requests, hparams, cache_name, stats_dir, False, False, verbose, stage1)
    if verbose:
        print(f"New weights successfully inserted into {[f'{n}.weight' for n in names]}")
    return pipe, orig_unet


# ---- SDXL ----------------------------------------------------------------------------------------------

def _sdxl_overrides(hparams, mom2_weight, mom2_weight_2, edit_weight):
    hparams.mom2_update_weight = mom2_weight if mom2_weight is not None else hparams.mom2_update_weight
    hparams.mom2_update_weight_2 = mom2_weight_2 if mom2_weight_2 is not None else hparams.mom2_update_weight_2
    hparams.edit_weight = edit_weight if edit_weight is not None else hparams.edit_weight


SDXL_TE1_COST_SHARE = 0.14     # TE1 alone 16 ms, TE2 alone 102 ms per 1 000 concepts on one MI355X: share of the ranks TE1 gets
_SDXL_GROUPS: Dict[tuple, tuple] = {}


def sdxl_rank_split(rank: int, world: int):
    """(encoder this rank edits, its rank inside that encoder's group, group sizes (g1, g2)).  The two SDXL text encoders
    are independent models (reference :1233-1320 vs :1333-1422): TE1 goes to the first g1 ranks, TE2 to the rest, in
    proportion to their cost (at least one rank each)."""
    g1 = max(1, min(world - 1, int(round(world * SDXL_TE1_COST_SHARE))))
    return (1, rank, (g1, world - g1)) if rank < g1 else (2, rank - g1, (g1, world - g1))


def _sdxl_split(shard: ConceptShard):
    """Process groups of the TE1 / TE2 rank split, or None when the split does not apply (one rank, a caller-supplied
    group, or EMCID_SDXL_SPLIT=0: then both encoders are edited on every rank, concept-sharded, on two streams)."""
    import torch.distributed as dist
    if shard.world < 2 or shard.group is not None or os.environ.get("EMCID_SDXL_SPLIT", "1") == "0":
        return None
    which, sub_rank, (g1, g2) = sdxl_rank_split(shard.rank, shard.world)
    key = (shard.world, g1)
    if key not in _SDXL_GROUPS:       # collective: every rank creates both groups, in the same order
        _SDXL_GROUPS[key] = (dist.new_group(list(range(g1))), dist.new_group(list(range(g1, shard.world))))
    grp = _SDXL_GROUPS[key][which - 1]
    return {"which": which, "shard": ConceptShard(sub_rank, g1 if which == 1 else g2, grp), "roots": (0, g1)}


def _axpy_weight_(w: torch.Tensor, dW: torch.Tensor):
    """``w += dW`` on an encoder weight (the kernel writes through the raw pointer) with the in-place version counter of the
    PARAMETER bumped: the caches derived from a weight (split-fp16 planes, native layer structs) follow that counter, and a
    write through ``w.data`` alone leaves it where it was (``.data`` has a counter of its own)."""
    hip.axpy_(w.data, dW)
    edit_engine._touch(w)


def _broadcast_(t: torch.Tensor, src: int):
    """In-place broadcast over the default group (RCCL on device buffers; gloo stages HBM tensors through the host).  ``t`` may
    be a Parameter: the bytes go through ``t.data`` and the parameter's version counter is bumped (see ``_axpy_weight_``)."""
    import torch.distributed as dist
    raw = t.data
    if raw.is_cuda and dist.get_backend() == "gloo":
        host = raw.cpu()
        dist.broadcast(host, src=src)
        raw.copy_(host)
    else:
        dist.broadcast(raw, src=src)
    edit_engine._touch(t)


def _sdxl_plans(pipe, requests, hparams, cache_name, stat_dir, stat_dir_2, verbose, shard, stage1):
    if hparams.num_edit_tokens != 1:
        raise AssertionError("num_edit_tokens should be 1")   # reference :1246
    p1 = prepare_text_encoder_edit(pipe.text_encoder, pipe.tokenizer, requests, hparams, hparams.layers,
                                   hparams.mom2_update_weight, stat_dir, cache_name, "", verbose, shard, stage1)
    p2 = prepare_text_encoder_edit(pipe.text_encoder_2, pipe.tokenizer_2, requests, hparams, hparams.layers_2,
                                   hparams.mom2_update_weight_2, stat_dir_2, cache_name, "_2", verbose, shard, stage1)
    return p1, p2


@_retry_if_stale
def execute_emcid_sd_xl_text_encoders(pipe, requests: List[Dict], hparams: EMCIDXLHyperParams,
                                      cache_name: Optional[str] = None, mom2_weight: Optional[int] = None,
                                      mom2_weight_2: Optional[int] = None, edit_weight: Optional[float] = None,
                                      verbose: bool = True, stat_dir="data/stats/sdxl/text1",
                                      stat_dir_2="data/stats/sdxl/text2", shard=None, stage1=None):
    """(deltas, deltas_2).  TE1 is restored; TE2 is left at W + dW exactly as the reference leaves it (:1410)."""
    _sdxl_overrides(hparams, mom2_weight, mom2_weight_2, edit_weight)
    _announce(requests, verbose)
    stage1 = _default_stage1_sdxl(pipe, hparams, stage1)
    p1, p2 = _sdxl_plans(pipe, requests, hparams, cache_name, stat_dir, stat_dir_2, verbose, shard, stage1)
    e1 = run_checked(p1, keep_factors=True, restore=True)
    e2 = run_checked(p2, keep_factors=True, restore=not SDXL_TE2_DOUBLE_APPLY)
    return _deltas_to_host(e1), _deltas_to_host(e2)


@_retry_if_stale
def apply_emcid_to_sdxl_text_encoders(pipe, requests: List[Dict], hparams: EMCIDXLHyperParams, device: str,
                                      mom2_weight: Optional[int] = None, mom2_weight_2: Optional[int] = None,
                                      edit_weight: Optional[float] = None, return_orig_text_encoder=False,
                                      cache_name: Optional[str] = None, stat_dir=XL_STATS_DIR1,
                                      stat_dir_2=XL_STATS_DIR2, verbose: bool = True, shard=None, stage1=None):
    """Returns (pipe, original text_encoder | None, original text_encoder_2 | None)."""
    o1 = deepcopy(pipe.text_encoder) if return_orig_text_encoder else None
    o2 = deepcopy(pipe.text_encoder_2) if return_orig_text_encoder else None
    _sdxl_overrides(hparams, mom2_weight, mom2_weight_2, edit_weight)
    _announce(requests, verbose)
    stage1 = _default_stage1_sdxl(pipe, hparams, stage1)
    split = _sdxl_split(_shard_from_env(shard))
    if split is not None:
        # TE1 || TE2 on disjoint GPU groups (BASELINE config 4): this rank edits ONE encoder, concept-sharded
        # inside its group, then the groups' roots broadcast the edited fc2 weights so every rank ends with the whole pipe
        if hparams.num_edit_tokens != 1:
            raise AssertionError("num_edit_tokens should be 1")   # reference :1246
        stale = None
        try:
            if split["which"] == 1:
                plan = prepare_text_encoder_edit(pipe.text_encoder, pipe.tokenizer, requests, hparams, hparams.layers,
                                                 hparams.mom2_update_weight, stat_dir, cache_name, "", verbose, split["shard"], stage1)
                run_checked(plan, keep_factors=False, restore=False)
            else:
                plan = prepare_text_encoder_edit(pipe.text_encoder_2, pipe.tokenizer_2, requests, hparams, hparams.layers_2,
                                                 hparams.mom2_update_weight_2, stat_dir_2, cache_name, "_2", verbose, split["shard"], stage1)
                edits = run_checked(plan, keep_factors=False, restore=False)
                if SDXL_TE2_DOUBLE_APPLY:   # execute left TE2 edited, apply adds the update once more (:93-99)
                    for e in edits:
                        _axpy_weight_(nethook.get_parameter(pipe.text_encoder_2, e.weight_name), e.dW)
        except clip_forward.StaleWeightCacheError as e:
            stale = e              # (this group's weights are back: every rank of a group reads the same flag)
        # one group's stale caches must not leave the other group waiting in the broadcasts below: every rank learns of it
        if _any_rank(stale is not None, next(pipe.text_encoder.parameters()).device):
            if stale is None:      # the sound group's encoder is edited (TE2 sits at W + 2 dW): put it back before everybody raises
                with torch.no_grad():
                    for l, w0 in (plan.backups or {}).items():
                        nethook.get_parameter(plan.text_encoder, plan.weight_name(l)).copy_(w0)
            raise stale if stale is not None else clip_forward.StaleWeightCacheError(
                "the other encoder's rank group ran on stale weight caches; this group's weights were restored")
        names = []
        for enc, layers, root in ((pipe.text_encoder, hparams.layers, split["roots"][0]),
                                  (pipe.text_encoder_2, hparams.layers_2, split["roots"][1])):
            for layer in layers:
                name = f"{hparams.rewrite_module_tmp.format(layer)}.weight"
                _broadcast_(nethook.get_parameter(enc, name), root)
                names.append(name)
        if verbose:
            print(f"New weights successfully inserted into {names}")
        return pipe, o1, o2
    p1, p2 = _sdxl_plans(pipe, requests, hparams, cache_name, stat_dir, stat_dir_2, verbose, shard, stage1)
    # The two encoders are independent models (:1233 vs :1333): two HIP streams on one GPU.  Measured for the 1 000-concept edit (scripts/bench_sdxl.py): 83.0 ms against 84.9 ms per apply call — TE2's
    # 26-layer prefix forward is 3/4 of the call and leaves little for TE1 to hide under.  With more ranks the encoders go to
    # disjoint rank groups instead (_sdxl_split above).
    dev2 = next(pipe.text_encoder_2.parameters()).device      # (v* may still be a pending upload: plan.zs_t is set lazily)
    two_streams = True
    s2 = torch.cuda.Stream(device=dev2) if two_streams else torch.cuda.current_stream(dev2)
    if two_streams:
        s2.wait_stream(torch.cuda.current_stream(dev2))
    e1 = run_encoder_edit(p1, keep_factors=False, restore=False)
    with torch.cuda.stream(s2):
        e2 = run_encoder_edit(p2, keep_factors=False, restore=False)
        if SDXL_TE2_DOUBLE_APPLY:   # execute left TE2 edited, apply adds the update once more (:93-99)
            for e in e2:
                _axpy_weight_(nethook.get_parameter(pipe.text_encoder_2, e.weight_name), e.dW)
    if two_streams:
        torch.cuda.current_stream(dev2).wait_stream(s2)

    def double_apply(edits):
        for e in edits:
            _axpy_weight_(nethook.get_parameter(pipe.text_encoder_2, e.weight_name), e.dW)

    stale = None
    for plan, redo in ((p1, None), (p2, double_apply if SDXL_TE2_DOUBLE_APPLY else None)):
        try:
            check_info(plan)                       # restores the encoder's weights if a factorization failed
        except clip_forward.StaleWeightCacheError as e:
            stale = e                              # (this encoder's weights are back; the other one's follow below)
        except FloatingPointError:
            if os.environ.get("EMCID_LU_FALLBACK", "1") == "0":
                raise
            plan.solver, plan.cov_factors = "lu", None          # the reference's own solver semantics (edit_engine.run_checked)
            again = run_encoder_edit(plan, keep_factors=False, restore=False)
            try:
                check_info(plan)
            except clip_forward.StaleWeightCacheError as e:     # (weights back; the other encoder's follow below)
                stale = e
                continue
            if redo is not None:
                redo(again)
    if stale is not None:      # one encoder ran on stale planes: put BOTH back (TE2 sits at W + 2 dW) and let the call be redone
        with torch.no_grad():
            for plan in (p1, p2):
                for l, w0 in (plan.backups or {}).items():
                    nethook.get_parameter(plan.text_encoder, plan.weight_name(l)).copy_(w0)
        raise stale
    if verbose:
        print(f"New weights successfully inserted into {[e.weight_name for e in e1 + e2]}")
    return pipe, o1, o2


# ---- edited-weight export (the reference never saves its edits) --------------------------------

def export_edited_weights(text_encoder, hparams, path, layers: Optional[Sequence[int]] = None) -> List[str]:
    """Write the (edited) fc2 weights of ``layers`` (default ``hparams.layers``) to a safetensors file keyed by
    the parameter names of ``hparams.rewrite_module_tmp``; returns the names written."""
    from safetensors.torch import save_file
    layers = list(hparams.layers if layers is None else layers)
    names = [f"{hparams.rewrite_module_tmp.format(l)}.weight" for l in layers]
    tensors = {n: nethook.get_parameter(text_encoder, n).detach().to("cpu").contiguous() for n in names}
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    save_file(tensors, str(path), metadata={"format": "pt", "producer": "emcid_amd"})
    return names


def load_edited_weights(text_encoder, path) -> List[str]:
    """Copy the tensors of an ``export_edited_weights`` file into the encoder (shape-checked, in place)."""
    from safetensors.torch import load_file
    tensors = load_file(str(path))
    with torch.no_grad():
        for n, t in tensors.items():
            w = nethook.get_parameter(text_encoder, n)
            if w.shape != t.shape:
                raise ValueError(f"{n}: file has {tuple(t.shape)}, model has {tuple(w.shape)}")
            w.copy_(t.to(w.device, w.dtype))
    return list(tensors)


def apply_emcid_to_model(pipe, requests, hparams, device, **kwargs):
    """Dispatching alias (the entry-point name used by BASELINE.json; absent from the reference)."""
    if isinstance(hparams, EMCIDXLHyperParams) or hasattr(hparams, "layers_2"):
        return apply_emcid_to_sdxl_text_encoders(pipe, requests, hparams, device, **kwargs)
    return apply_emcid_to_text_encoder(pipe, requests, hparams, device, **kwargs)
