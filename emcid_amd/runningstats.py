"""Running second moment on the MI355X + the npz cache protocol around it.

Host-side counterpart of the slice of the reference's util/runningstats.py that the edit path uses:
``SecondMoment`` (:469-511), ``CombinedStat`` (:1347-1388), ``tally`` (:54-121), the npz cache I/O with
NaN-boxed ``None`` (:1409-1512) and the deterministic subset samplers (:1515-1603).

What differs by design (MI355X-first):
* ``SecondMoment.add`` does not call ``a.t().mm(a)``.  Batches are appended to a token staging buffer in
  HBM and flushed through the hand-written SYRK kernel (csrc/gram_f32.hip) once enough tokens are queued,
  so one launch sees a long contraction dimension and only the lower triangle is ever computed; the
  mirror happens when the statistic is read.
* ``all_reduce_`` sums the statistic over ``torch.distributed`` ranks (RCCL on the GPU box) for the
  caption-sharded Stage 0; the reference is single-process.
The on-disk npz written by ``tally`` is byte-compatible with the reference's (keys ``mom2.constructor``,
``mom2.count``, ``mom2.mom2``, ``sample_size``).
"""
import os
import random
import struct
from typing import Optional

import numpy
import torch
from torch.utils.data.sampler import Sampler

from . import hip


class Stat:
    def __init__(self, state):
        self.load_state_dict(resolve_state_dict(state))

    def add(self, x, *args, **kwargs):
        raise NotImplementedError

    def load_state_dict(self, d):
        raise NotImplementedError

    def state_dict(self):
        raise NotImplementedError

    def to_(self, device):
        pass

    def save(self, filename):
        save_cached_state(filename, self, {})

    def load(self, filename):
        self.load_state_dict(load_cached_state(filename, {}, quiet=True, throw=True))

    def _normalize_add_shape(self, x):
        if x.dim() != 2:
            x = x.reshape(-1, x.shape[-1])
        return x


class SecondMoment(Stat):
    """Non-centred second moment E[x x^T] accumulated in HBM.

    ``stage_tokens`` is the capacity of the staging buffer (tokens queued before a SYRK launch);
    ``ksplit`` is passed to the kernel (0 = choose, 1 = deterministic single pass)."""

    CONSTRUCTOR = "util.runningstats.SecondMoment()"  # what the reference's state_dict records (:504)

    def __init__(self, split_batch=True, state=None, stage_tokens: int = 16384, ksplit: int = 0):
        self.count = 0
        self._lower = None      # (d, d) device accumulator, lower triangle valid
        self._full = None       # cached mirrored matrix (any device), invalidated by add()
        self._stage = None
        self._staged = 0
        self.stage_tokens = stage_tokens
        self.ksplit = ksplit
        self.store_dtype = None   # dtype of the stored sums when it differs from the accumulator's (float16 statistics)
        self.split_batch = split_batch
        if state is not None:
            super().__init__(state)

    # -- accumulation --------------------------------------------------------------------------------
    def add(self, a: torch.Tensor, count: Optional[int] = None, row_weight: Optional[torch.Tensor] = None):
        """mom2 += a^T a; ``count`` (default: the rows of ``a``) is what the rows stand for — the packed Stage-0 forward
        hands in each distinct prefix once, scaled by the square root of its multiplicity: either already multiplied in, or as
        ``row_weight`` (rows,) — row r then enters as fl(row_weight[r] * a[r]), the product formed inside the Gram's own
        kernels when the batch goes straight to the split-fp16 path, and here otherwise."""
        a = self._normalize_add_shape(a)
        if len(a) == 0:
            return
        if row_weight is not None:
            direct = (a.is_cuda and a.dtype == torch.float32 and a.shape[0] >= self.stage_tokens and a.is_contiguous()
                      and row_weight.dtype == torch.float32 and hip.gram_takes_row_weight(a, self.ksplit)
                      and (self._lower is None or self._lower.dtype == torch.float32))
            if not direct:
                a, row_weight = a * row_weight.to(a.dtype).reshape(-1, 1), None
        if not a.is_cuda:
            raise hip.EmcidHipError("SecondMoment.add needs a tensor in HBM (no CPU path in emcid_amd)")
        if a.dtype not in (torch.float32, torch.float64):
            raise hip.EmcidHipError(f"SecondMoment.add accumulates fp32 (the reference's CLI default, fp32 MFMA SYRK) or fp64 "
                                    f"(its --precision float64, fp64 MFMA SYRK); got {a.dtype}")
        d = a.shape[1]
        if self._lower is None:
            if self._full is not None:   # resumed from a loaded state
                self._lower = self._full.to(a.device, a.dtype).contiguous().clone()
            else:
                self._lower = torch.zeros(d, d, dtype=a.dtype, device=a.device)
        if a.dtype != self._lower.dtype:
            raise hip.EmcidHipError(f"SecondMoment holds {self._lower.dtype} sums; got a {a.dtype} batch")
        if a.dtype == torch.float64 and d % 2:
            raise hip.EmcidHipError("fp64 statistics need an even feature width (16-byte aligned f64 rows)")
        self._full = None
        self.count += a.shape[0] if count is None else int(count)
        if a.shape[0] >= self.stage_tokens:
            self.flush()
            self._accumulate(a.contiguous(), None if row_weight is None else row_weight.reshape(-1).contiguous())
            return
        if self._stage is None:
            self._stage = torch.empty(self.stage_tokens, d, dtype=a.dtype, device=a.device)
        if self._staged + a.shape[0] > self.stage_tokens:
            self.flush()
        self._stage[self._staged:self._staged + a.shape[0]].copy_(a)
        self._staged += a.shape[0]

    def _accumulate(self, x: torch.Tensor, row_weight: Optional[torch.Tensor] = None):
        """lower(mom2) += x^T x on the matrix cores: fp32 SYRK (csrc/gram_f32.hip) or the fp64 MFMA GEMM restricted to the
        lower tiles (csrc/gemm_f64.h: A = x stored [K = tokens][rows = d] for both operands)."""
        if x.dtype == torch.float32:
            hip.gram_accumulate_(self._lower, x, self.ksplit, row_weight=row_weight)
        else:
            assert row_weight is None
            hip.dgemm_ex(1, 1, x, x, self._lower, alpha=1.0, beta=1.0, flags=16, ksplit=1 if self.ksplit == 1 else 0)

    def flush(self):
        if self._staged:
            self._accumulate(self._stage[:self._staged])
            self._staged = 0

    # -- readout -------------------------------------------------------------------------------------
    @property
    def mom2(self) -> Optional[torch.Tensor]:
        if self._full is None and self._lower is not None:
            self.flush()
            if self._lower.dtype == torch.float32:
                self._full = hip.symmetrize_lower_(self._lower.clone())
            else:
                self._full = torch.tril(self._lower) + torch.tril(self._lower, -1).t()
        return self._full

    def moment(self):
        return self.mom2 / self.count

    def to_(self, device):
        if self._lower is not None or self._full is not None:
            full = self.mom2.to(device)
            self._full, self._lower, self._stage, self._staged = full, None, None, 0

    def all_reduce_(self, group=None):
        """Sum over ranks (caption-sharded Stage 0): mom2 with one all-reduce, count with another."""
        import torch.distributed as dist

        self.flush()
        acc = self._lower if self._lower is not None else self._full
        # An empty shard (a rank that was dealt no caption) must not leave its peers waiting inside the all-reduce below: every
        # rank first learns whether ALL ranks hold data (one tiny MIN all-reduce), and all of them raise the same error if not.
        backend = dist.get_backend(group)
        flag_dev = "cpu" if backend == "gloo" else (acc.device if acc is not None and acc.is_cuda
                                                    else torch.device("cuda", torch.cuda.current_device()))
        have = torch.tensor([0 if acc is None else 1], dtype=torch.int32, device=flag_dev)
        dist.all_reduce(have, op=dist.ReduceOp.MIN, group=group)
        if int(have.item()) == 0:
            raise RuntimeError("all_reduce_ on an empty SecondMoment: at least one rank of the group collected nothing "
                               "(sample smaller than the world size?) — raised on every rank")
        staged = acc.is_cuda and backend == "gloo"   # gloo (tests) cannot take HBM tensors; RCCL does
        if staged:
            host = acc.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            acc.copy_(host)
        else:
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
        cnt = torch.tensor([self.count], dtype=torch.int64, device="cpu" if staged else acc.device)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
        self.count = int(cnt.item())
        if self._lower is not None:
            self._full = None

    def state_dict(self):
        m2 = self.mom2.cpu()
        if self.store_dtype is not None and m2.dtype != self.store_dtype:
            m2 = m2.to(self.store_dtype)
        return dict(constructor=self.CONSTRUCTOR, count=self.count, mom2=m2.numpy())

    def load_state_dict(self, state):
        self.count = int(state["count"])
        self._full = torch.from_numpy(numpy.asarray(state["mom2"]))
        self._lower, self._stage, self._staged = None, None, 0


class Mean(Stat):
    """Running mean of the rows fed to ``add`` (reference: util/runningstats.py:234-291, Chan-style batch update) with the
    reference's npz keys (count, batchcount, data_shape, mean).  Stage 1's EWC term reads the Fisher diagonal that
    emcid/fim_cal.py stored through this class (emcid/compute_z.py:478-486); host-side torch, no kernel."""

    def __init__(self, state=None):
        if state is not None:
            return super().__init__(state)
        self.count = 0
        self.batchcount = 0
        self._mean = None
        self.data_shape = None

    def add(self, a):
        a = self._normalize_add_shape(a)
        if len(a) == 0:
            return
        batch_count = a.shape[0]
        batch_mean = a.sum(0) / batch_count
        self.batchcount += 1
        if self._mean is None:
            self.count = batch_count
            self._mean = batch_mean
            return
        self.count += batch_count
        self._mean.add_(batch_mean.sub_(self._mean).mul_(float(batch_count) / self.count))

    def size(self):
        return self.count

    def mean(self):
        return self._mean

    def to_(self, device):
        if self._mean is not None:
            self._mean = self._mean.to(device)

    def load_state_dict(self, state):
        self.count = int(state["count"])
        self.batchcount = int(state["batchcount"])
        self._mean = torch.from_numpy(numpy.asarray(state["mean"]))
        ds = state["data_shape"]
        if ds is None or is_null_numpy_value(ds) or (isinstance(ds, numpy.ndarray) and ds.dtype == object and ds.item() is None):
            self.data_shape = None
        else:
            self.data_shape = tuple(int(v) for v in numpy.asarray(ds).reshape(-1))

    def state_dict(self):
        return dict(constructor="util.runningstats.Mean()", count=self.count,
                    data_shape=self.data_shape and tuple(self.data_shape), batchcount=self.batchcount,
                    mean=self._mean.cpu().numpy())


class CombinedStat(Stat):
    """Bundle of named stats sharing one ``add`` / one npz (keys are ``<name>.<key>``)."""

    def __init__(self, state=None, **kwargs):
        self._objs = kwargs
        if state is not None:
            super().__init__(state)

    def __getattr__(self, k):
        objs = self.__dict__.get("_objs", {})
        if k in objs:
            return objs[k]
        raise AttributeError(k)

    def add(self, d, *args, row_weight=None, **kwargs):
        scaled = None
        for obj in self._objs.values():
            if row_weight is None:
                obj.add(d, *args, **kwargs)
            elif isinstance(obj, SecondMoment):
                obj.add(d, *args, row_weight=row_weight, **kwargs)      # applied inside the Gram's kernels where it can be
            else:
                if scaled is None:
                    scaled = d * row_weight.to(d.dtype).reshape(-1, 1)
                obj.add(scaled, *args, **kwargs)

    def load_state_dict(self, state):
        for prefix, obj in self._objs.items():
            obj.load_state_dict(pull_key_prefix(prefix, state))

    def state_dict(self):
        out = {}
        for prefix, obj in self._objs.items():
            out.update(push_key_prefix(prefix, obj.state_dict()))
        return out

    def to_(self, device):
        for v in self._objs.values():
            v.to_(device)

    def all_reduce_(self, group=None):
        for v in self._objs.values():
            v.all_reduce_(group)


def push_key_prefix(prefix, d):
    return {f"{prefix}.{k}": v for k, v in d.items()}


def pull_key_prefix(prefix, d):
    head = prefix + "."
    keys = d.files if hasattr(d, "files") else d.keys()
    return {k[len(head):]: d[k] for k in keys if k.startswith(head)}


# ---- npz cache protocol (None is stored as the NaN with payload 0xfff8000000000002) -----------------------

_NULL_BITS = 0xFFF8000000000002
null_numpy_value = numpy.array(struct.unpack(">d", struct.pack(">Q", _NULL_BITS))[0], dtype=numpy.float64)


def is_null_numpy_value(v):
    return (isinstance(v, numpy.ndarray) and numpy.ndim(v) == 0 and v.dtype == numpy.float64 and numpy.isnan(v)
            and struct.unpack(">Q", struct.pack(">d", float(v)))[0] == _NULL_BITS)


def box_numpy_null(d):
    if isinstance(d, dict):
        return {k: box_numpy_null(v) for k, v in d.items()}
    return null_numpy_value if d is None else d


def unbox_numpy_null(d):
    if hasattr(d, "files"):
        return {k: unbox_numpy_null(d[k]) for k in d.files}
    if isinstance(d, dict):
        return {k: unbox_numpy_null(v) for k, v in d.items()}
    return None if is_null_numpy_value(d) else d


def read_npz_stored(path, alloc=None) -> Optional[dict]:
    """{member: array} of an npz whose members are all STORED (uncompressed — what ``numpy.savez`` writes, hence every statistics
    file of the reference, util/runningstats.py:1409-1454) with plain numeric / string dtypes, parsed straight from the bytes: the
    local file headers (zip64 sizes from their extra field), each member's npy header, a ``frombuffer`` view of its data.
    ``numpy.load`` goes through ``zipfile``, which computes a CRC-32 over every member it reads — 62 % of the 35 ms a 37.7 MB
    second-moment file costs to load.  Returns None for anything else (compressed or pickled members, data descriptors, other
    layouts): the caller then uses ``numpy.load`` and gets numpy's behaviour, errors included.  ``alloc(nbytes)``: where the
    file's bytes go — a writable uint8 array of that size (e.g. the numpy view of a page-locked torch tensor, so that a member
    can be uploaded without a staging copy); default: a fresh numpy array."""
    import ast
    try:
        if alloc is None:
            buf = numpy.fromfile(path, dtype=numpy.uint8)  # writable: the arrays below are views of it (torch wants writable ones)
        else:
            with open(path, "rb", buffering=0) as f:
                size = os.fstat(f.fileno()).st_size
                buf = alloc(size)
                got, view = 0, memoryview(buf)
                while got < size:
                    r = f.readinto(view[got:])
                    if not r:
                        return None
                    got += r
    except (OSError, ValueError):
        return None
    blob = memoryview(buf)
    out, o, n = {}, 0, len(blob)
    try:
        while o + 30 <= n and blob[o:o + 4] == b"PK\x03\x04":
            flags, method = int.from_bytes(blob[o + 6:o + 8], "little"), int.from_bytes(blob[o + 8:o + 10], "little")
            csize, usize = int.from_bytes(blob[o + 18:o + 22], "little"), int.from_bytes(blob[o + 22:o + 26], "little")
            n_name, n_extra = int.from_bytes(blob[o + 26:o + 28], "little"), int.from_bytes(blob[o + 28:o + 30], "little")
            if method != 0 or (flags & 0x08):              # compressed, or sizes in a trailing data descriptor
                return None
            name = bytes(blob[o + 30:o + 30 + n_name]).decode("utf-8")
            extra = bytes(blob[o + 30 + n_name:o + 30 + n_name + n_extra])
            e = 0
            while e + 4 <= len(extra):                     # zip64 extended information: the real 8-byte sizes
                hid, hlen = int.from_bytes(extra[e:e + 2], "little"), int.from_bytes(extra[e + 2:e + 4], "little")
                if hid == 0x0001 and hlen >= 16:
                    usize, csize = int.from_bytes(extra[e + 4:e + 12], "little"), int.from_bytes(extra[e + 12:e + 20], "little")
                e += 4 + hlen
            d0 = o + 30 + n_name + n_extra
            if csize != usize or csize == 0xFFFFFFFF or d0 + csize > n or not name.endswith(".npy"):
                return None
            if blob[d0:d0 + 6] != b"\x93NUMPY":
                return None
            major = blob[d0 + 6]
            if major == 1:
                hlen, h0 = int.from_bytes(blob[d0 + 8:d0 + 10], "little"), d0 + 10
            elif major in (2, 3):
                hlen, h0 = int.from_bytes(blob[d0 + 8:d0 + 12], "little"), d0 + 12
            else:
                return None
            meta = ast.literal_eval(bytes(blob[h0:h0 + hlen]).decode("latin1"))
            dt = numpy.dtype(meta["descr"])
            shape = tuple(meta["shape"])
            if dt.hasobject or (meta["fortran_order"] and len(shape) > 1):
                return None
            count = int(numpy.prod(shape, dtype=numpy.int64)) if shape else 1
            p0 = h0 + hlen
            if p0 + count * dt.itemsize > d0 + csize:
                return None
            out[name[:-4]] = numpy.frombuffer(buf, dtype=dt, count=count, offset=p0).reshape(shape)
            o = d0 + csize
        if not out or blob[o:o + 4] != b"PK\x01\x02":      # the central directory must follow the last member
            return None
    except Exception:
        return None
    return out


def resolve_state_dict(s):
    if isinstance(s, (str, os.PathLike)):
        with numpy.load(s) as z:
            return unbox_numpy_null(z)
    return s


global_load_cache_enabled = True


class cache_load_enabled:
    """``with cache_load_enabled(False):`` forces recomputation (reference :124-142)."""

    def __init__(self, enabled=True):
        self.prev, self.enabled = None, enabled

    def __enter__(self):
        global global_load_cache_enabled
        self.prev, global_load_cache_enabled = global_load_cache_enabled, self.enabled

    def __exit__(self, *a):
        global global_load_cache_enabled
        global_load_cache_enabled = self.prev


def load_cached_state(cachefile, args, quiet=False, throw=False):
    """None when caching is off, the file is absent/unreadable, or a recorded arg (``sample_size``) differs."""
    if not global_load_cache_enabled or cachefile is None:
        return None
    try:
        if isinstance(cachefile, dict):
            dat, label = cachefile, "state"
        else:
            dat = read_npz_stored(cachefile)               # the reference's own files, without zipfile's CRC pass
            if dat is not None:
                dat = unbox_numpy_null(dat)
            else:
                with numpy.load(cachefile) as z:
                    dat = unbox_numpy_null(z)
            label = cachefile
        for k, v in args.items():
            if k not in dat or dat[k] != v:
                if not quiet:
                    print(f"{label} {k} changed from {dat.get(k)} to {v}")
                return None
    except (FileNotFoundError, ValueError) as e:
        if throw:
            raise e
        return None
    if not quiet:
        print(f"Loading cached {label}")
    return dat


def save_cached_state(cachefile, obj, args):
    if cachefile is None:
        return
    dat = obj.state_dict()
    for k, v in args.items():
        if k in dat:
            assert dat[k] == v
        dat[k] = v
    if isinstance(cachefile, dict):
        cachefile.clear()
        cachefile.update(dat)
    else:
        os.makedirs(os.path.dirname(str(cachefile)) or ".", exist_ok=True)
        numpy.savez(cachefile, **box_numpy_null(dat))


# ---- deterministic subset sampling ----------------------------------------------------------------------

class FixedSubsetSampler(Sampler):
    def __init__(self, samples):
        self.samples = list(samples)

    def __iter__(self):
        return iter(self.samples)

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, key):
        return self.samples[key]

    def dereference(self, indices):
        return [self.samples[i] for i in indices]

    def subset(self, new_subset):
        return FixedSubsetSampler(self.dereference(new_subset))

    def shard(self, rank: int, world: int):
        """Contiguous 1/world slice of the SAME fixed sample (partitioned, never re-drawn)."""
        n = len(self.samples)
        lo, hi = (n * rank) // world, (n * (rank + 1)) // world
        return FixedSubsetSampler(self.samples[lo:hi])


class FixedRandomSubsetSampler(FixedSubsetSampler):
    """``random.Random(seed).shuffle(range(len(ds)))[start:end]`` — the reference's fixed pseudo-random sample."""

    def __init__(self, data_source, start=None, end=None, seed=1):
        order = list(range(len(data_source)))
        random.Random(seed).shuffle(order)
        self.data_source = data_source
        super().__init__(order[start:end])


def make_sampler(dataset, sample_size=None, sampler=None, random_sample=None, shard=None):
    """The fixed sample subset ``make_loader`` would iterate (same order), as a FixedSubsetSampler."""
    if sample_size is not None:
        assert sampler is None, "sampler cannot be specified with sample_size"
        sample_size = min(sample_size, len(dataset))
        if random_sample is None:
            sampler = FixedSubsetSampler(range(sample_size))
        else:
            sampler = FixedRandomSubsetSampler(dataset, seed=random_sample, end=sample_size)
    if sampler is None:
        sampler = FixedSubsetSampler(range(len(dataset)))
    if shard is not None:
        sampler = sampler.shard(*shard)
    return sampler


def make_loader(dataset, sample_size=None, batch_size=1, sampler=None, random_sample=None, shard=None, **kwargs):
    """DataLoader over a fixed sample subset.  ``shard=(rank, world)`` keeps this rank's slice of it."""
    if callable(dataset) and not hasattr(dataset, "__len__"):
        dataset = dataset()
    if isinstance(dataset, torch.Tensor):
        dataset = torch.utils.data.TensorDataset(dataset)
    if sample_size is not None:
        assert sampler is None, "sampler cannot be specified with sample_size"
        if sample_size > len(dataset):
            print(f"Warning: sample size {sample_size} > dataset size {len(dataset)}")
            sample_size = len(dataset)
        if random_sample is None:
            sampler = FixedSubsetSampler(range(sample_size))
        else:
            sampler = FixedRandomSubsetSampler(dataset, seed=random_sample, end=sample_size)
    if shard is not None:
        if sampler is None:
            sampler = FixedSubsetSampler(range(len(dataset)))
        sampler = sampler.shard(*shard)
    return torch.utils.data.DataLoader(dataset, sampler=sampler, batch_size=batch_size, **kwargs)


def tally(stat, dataset, cache=None, quiet=False, shard=None, group=None, **kwargs):
    """Loader whose exhaustion finalises ``stat``: (all-reduce over ranks if sharded) -> cpu -> npz cache.
    If the cache already holds the statistic (same ``sample_size``) the stat is loaded and the loader is empty."""
    assert isinstance(stat, Stat)
    args = {k: kwargs[k] for k in ("sample_size",) if k in kwargs}
    cached = load_cached_state(cache, args, quiet=quiet)
    if cached is not None:
        stat.load_state_dict(cached)
        return iter(())
    loader = make_loader(dataset, shard=shard, **kwargs)

    def wrapped():
        yield from loader
        if shard is not None and shard[1] > 1:
            stat.all_reduce_(group)
        stat.to_(device="cpu")
        if cache is not None and (shard is None or shard[0] == 0):
            save_cached_state(cache, stat, args)

    return wrapped()
