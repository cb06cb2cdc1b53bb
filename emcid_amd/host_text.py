"""ctypes binding of ``libemcid_host.so`` (``include/emcid_host.h``): CLIP byte-level BPE encoding and the subject
token-range walk in C++ for the prompts of an edit.

The reference tokenizes with the pipeline's Hugging Face tokenizer (``emcid/compute_z.py:65``) and searches with
``find_token_range`` (``experiments/causal_trace.py:1057``).  ``NativeClipBpe.for_tokenizer`` reads the HF tokenizer's own
serialized configuration, accepts it only if it is exactly the CLIP pipeline the library restates (normalizer, pre-tokenizer,
BPE options, post-processor), builds the native model from the HF vocabulary and merges, and checks a probe set against the HF
tokenizer; any difference disables the native path for that tokenizer.  Prompts the library does not serve (non-ASCII, special
-token syntax, characters outside the vocabulary) come back flagged and go through the HF tokenizer, row by row."""
import ctypes
import json
import os
import threading
import weakref
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np

ABI_VERSION = 5
_LIB = None
_LOCK = threading.Lock()

_CLIP_SPLIT = r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"


def lib_path() -> Path:
    return Path(__file__).resolve().parent / "csrc" / "libemcid_host.so"


def load():
    global _LIB
    with _LOCK:
        if _LIB is not None:
            return _LIB
        path = lib_path()
        if not path.exists():
            raise RuntimeError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = ctypes.CDLL(str(path))
        P, I64, I32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
        lib.emcid_host_abi_version.restype = ctypes.c_int
        lib.emcid_host_last_error.restype = ctypes.c_char_p
        lib.emcid_bpe_create.restype = P
        lib.emcid_bpe_create.argtypes = [P, P, P, I64, P, I64, ctypes.c_char_p]
        lib.emcid_bpe_destroy.argtypes = [P]
        lib.emcid_bpe_destroy.restype = None
        lib.emcid_bpe_encode_batch.restype = I64
        lib.emcid_bpe_encode_batch.argtypes = [P, P, P, I64, I32, I32, I32, I32, P, P, P]
        lib.emcid_find_token_ranges.restype = I64
        lib.emcid_find_token_ranges.argtypes = [P, I64, I64, P, P, P, I64, P, P, ctypes.c_char_p, P, P, P]
        lib.emcid_bpe_encode_templated.restype = I64
        lib.emcid_bpe_encode_templated.argtypes = [P, P, P, P, P, I64, P, P, I64, P, P, I64, I32, I32, I32, I32, P, P, P, P]
        lib.emcid_find_token_ranges_idx.restype = I64
        lib.emcid_find_token_ranges_idx.argtypes = [P, I64, I64, P, P, P, I64, P, P, P, I64, ctypes.c_int, ctypes.c_char_p, P, P, P]
        lib.emcid_trie_build.restype = P
        lib.emcid_trie_build.argtypes = [P, I64, I64, P, I64]
        lib.emcid_trie_sizes.restype = None
        lib.emcid_trie_sizes.argtypes = [P, P, P, P, P]
        lib.emcid_trie_packed_bytes.restype = I64
        lib.emcid_trie_packed_bytes.argtypes = [P]
        lib.emcid_trie_export.restype = ctypes.c_int
        lib.emcid_trie_export.argtypes = [P, P, I64]
        lib.emcid_trie_destroy.restype = None
        lib.emcid_trie_destroy.argtypes = [P]
        lib.emcid_read_npz_rows_f32.restype = I64
        lib.emcid_read_npz_rows_f32.argtypes = [ctypes.c_char_p, P, I64, ctypes.c_char_p, I64, P, I64, P, I32]
        if lib.emcid_host_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{path}: ABI {lib.emcid_host_abi_version()}, this package needs {ABI_VERSION}; rebuild")
        _LIB = lib
        return lib


def available() -> bool:
    if os.environ.get("EMCID_NATIVE_TEXT", "1") == "0":
        return False
    try:
        load()
        return True
    except (RuntimeError, OSError):
        return False


def _ptr(a: np.ndarray):
    return ctypes.c_void_p(a.ctypes.data)


def pack_strings(items: Sequence[str]):
    """(bytes, int64 offsets) of the UTF-8 encodings of ``items`` laid end to end."""
    blob = "".join(items)
    if blob.isascii():
        off = np.zeros(len(items) + 1, dtype=np.int64)
        np.cumsum(np.fromiter(map(len, items), dtype=np.int64, count=len(items)), out=off[1:])
        return blob.encode("ascii"), off
    enc = [s.encode("utf-8", "surrogatepass") for s in items]
    off = np.zeros(len(items) + 1, dtype=np.int64)
    np.cumsum(np.fromiter(map(len, enc), dtype=np.int64, count=len(enc)), out=off[1:])
    return b"".join(enc), off


def _clip_pipeline(cfg: Dict) -> Optional[str]:
    """None if the serialized tokenizer is exactly the CLIP pipeline libemcid_host restates, else what differs."""
    m = cfg.get("model") or {}
    if m.get("type") != "BPE":
        return "model is not BPE"
    if m.get("dropout") or m.get("continuing_subword_prefix") or m.get("fuse_unk") or m.get("byte_fallback") \
            or m.get("ignore_merges"):
        return "BPE options"
    if not m.get("end_of_word_suffix"):
        return "no end-of-word suffix"
    nz = cfg.get("normalizer") or {}
    want_nz = [{"type": "NFC"}, {"type": "Replace", "pattern": {"Regex": r"\s+"}, "content": " "}, {"type": "Lowercase"}]
    if nz.get("type") != "Sequence" or nz.get("normalizers") != want_nz:
        return "normalizer"
    pt = cfg.get("pre_tokenizer") or {}
    pts = pt.get("pretokenizers") or []
    if pt.get("type") != "Sequence" or len(pts) != 2:
        return "pre-tokenizer"
    sp, bl = pts
    if sp.get("type") != "Split" or sp.get("pattern") != {"Regex": _CLIP_SPLIT} or sp.get("behavior") != "Removed" \
            or sp.get("invert") is not True:
        return "split pattern"
    if bl.get("type") != "ByteLevel" or bl.get("add_prefix_space") is not False:
        return "byte-level pre-tokenizer"
    pp = cfg.get("post_processor") or {}
    if pp.get("type") != "RobertaProcessing" or pp.get("add_prefix_space") not in (False, None):
        return "post-processor"
    for t in cfg.get("added_tokens") or []:
        if "<|" not in t.get("content", ""):
            return f"added token {t.get('content')!r}"
    return None


_PROBE_WORDS = ("a photo of", "An Image of the", "don't", "it's we're they've i'm she'll he'd", "rock'n'roll", "'tis", "''s",
                "x1y22z 007", "hello,world!!", "(a)[b]{c}", "tabs\tand\nnewlines\r\n", "  leading and trailing  ",
                "UPPER lower MiXeD", "semi;colon: dash-dash -- under_score", "a.b.c...", "100% #1 @home $5 & more *", "'", "''",
                "q'", "'re", "'l", "", " ", "~`^|\\/<>?=+", "the quick brown fox jumps over the lazy dog " * 12)


class NativeClipBpe:
    """The native twin of one HF CLIP tokenizer (``tokenizer._tokenizer`` is a ``tokenizers.Tokenizer``)."""

    def __init__(self, tokenizer, cfg: Dict):
        lib = load()
        model = cfg["model"]
        vocab: Dict[str, int] = model["vocab"]
        toks = list(vocab.keys())
        blob, off = pack_strings(toks)
        ids = np.fromiter((vocab[t] for t in toks), dtype=np.int32, count=len(toks))
        pairs = []
        for mg in model["merges"]:
            a, b = mg.split(" ") if isinstance(mg, str) else mg
            pairs.append((vocab[a], vocab[b]))
        merges = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
        self._lib = lib
        self._h = lib.emcid_bpe_create(blob, _ptr(off), _ptr(ids), len(toks), _ptr(merges) if len(merges) else None,
                                       len(merges), model["end_of_word_suffix"].encode())
        if not self._h:
            raise RuntimeError((lib.emcid_host_last_error() or b"").decode())
        weakref.finalize(self, lib.emcid_bpe_destroy, self._h)
        pp = cfg["post_processor"]
        self.bos, self.eos = int(pp["cls"][1]), int(pp["sep"][1])
        self.pad = int(tokenizer.pad_token_id)
        self.max_len = int(tokenizer.model_max_length)
        self.signature = self._signature(tokenizer)

    @staticmethod
    def _signature(tokenizer):
        return (id(tokenizer._tokenizer), len(tokenizer), tokenizer.pad_token_id, tokenizer.model_max_length,
                getattr(tokenizer, "padding_side", "right"), getattr(tokenizer, "truncation_side", "right"))

    def encode(self, prompts: Sequence[str]):
        """(ids (B, max_len) int64 padded with the pad id, lengths (B,) int32, fallback (B,) bool)."""
        n = len(prompts)
        blob, off = pack_strings(prompts)
        ids = np.empty((n, self.max_len), dtype=np.int64)
        lengths = np.empty(n, dtype=np.int32)
        fb = np.empty(n, dtype=np.uint8)
        rc = self._lib.emcid_bpe_encode_batch(self._h, blob, _ptr(off), n, self.bos, self.eos, self.pad, self.max_len,
                                              _ptr(ids), _ptr(lengths), _ptr(fb))
        if rc < 0:
            raise RuntimeError((self._lib.emcid_host_last_error() or b"").decode())
        return ids, lengths, fb.astype(bool)

    def encode_templated(self, pre: Sequence[str], suf: Sequence[str], names: Sequence[str], tmpl_idx: np.ndarray,
                         name_idx: np.ndarray, want_name_last: bool = False, narrow: bool = False):
        """``encode`` of the prompts ``pre[t] + names[k] + suf[t]`` for (t, k) = (tmpl_idx[i], name_idx[i]) without building
        the strings: (ids (B, max_len), lengths, fallback) exactly as ``encode`` gives for them.  ``want_name_last``: a fourth
        array, the position of the name's last token in each row (-1 where the row is not a plain concatenation).  ``narrow``:
        ids (B, W) with W = min(max_len, longest prompt's characters + 2) — the same rows without the padding columns nothing
        can reach."""
        n = len(tmpl_idx)
        pb, po = pack_strings(pre)
        sb, so = pack_strings(suf)
        nb, no = names if isinstance(names, tuple) else pack_strings(names)      # (bytes, offsets) from pack_strings, or strings
        tmpl_idx = np.ascontiguousarray(tmpl_idx, dtype=np.int32)
        name_idx = np.ascontiguousarray(name_idx, dtype=np.int32)
        # a prompt has at most one token per character: rows as wide as the longest prompt's characters + BOS / EOS hold every
        # row whole (same ids, lengths and cuts as model_max_length-wide rows; 3 000 rows of 77 are 1.8 MB of padding to write)
        width = self.max_len
        if narrow and len(no) > 1 and len(pre):
            chars = int(np.diff(po).max()) + int(np.diff(no).max()) + int(np.diff(so).max()) + 2
            width = max(2, min(self.max_len, chars))
        ids = np.empty((n, width), dtype=np.int64)
        lengths = np.empty(n, dtype=np.int32)
        fb = np.empty(n, dtype=np.uint8)
        name_last = np.empty(n, dtype=np.int32) if want_name_last else None
        rc = self._lib.emcid_bpe_encode_templated(self._h, pb, _ptr(po), sb, _ptr(so), len(pre), nb, _ptr(no), len(no) - 1,
                                                  _ptr(tmpl_idx), _ptr(name_idx), n, self.bos, self.eos, self.pad, width,
                                                  _ptr(ids), _ptr(lengths), _ptr(fb), _ptr(name_last) if want_name_last else None)
        if rc < 0:
            raise RuntimeError((self._lib.emcid_host_last_error() or b"").decode())
        if want_name_last:
            return ids, lengths, fb.astype(bool), name_last
        return ids, lengths, fb.astype(bool)

    def tokenize(self, tokenizer, prompts: Sequence[str]) -> Dict[str, np.ndarray]:
        """``tokenizer(prompts, padding=True, truncation=True)`` as (B, S) int64 arrays; flagged rows through ``tokenizer``."""
        ids, lengths, fb = self.encode(prompts)
        if fb.any():
            rows = np.nonzero(fb)[0].tolist()
            enc = tokenizer([prompts[i] for i in rows], padding=False, truncation=True)["input_ids"]
            for i, r in zip(rows, enc):
                ids[i, :len(r)] = r
                lengths[i] = len(r)
        S = int(lengths.max()) if len(prompts) else 0
        ids = np.ascontiguousarray(ids[:, :S])
        mask = (np.arange(S, dtype=np.int32)[None, :] < lengths[:, None]).astype(np.int64)
        return {"input_ids": ids, "attention_mask": mask}

    # ---- construction ------------------------------------------------------------------------------------------------------
    _BY_TOKENIZER = weakref.WeakKeyDictionary()

    @classmethod
    def for_tokenizer(cls, tokenizer) -> Optional["NativeClipBpe"]:
        """The checked native twin of ``tokenizer``, or None (not a CLIP tokenizers-backed tokenizer, library missing, probe
        mismatch).  Cached per tokenizer object; rebuilt if the tokenizer's vocabulary size or padding set-up changes."""
        if not available():
            return None
        bt = getattr(tokenizer, "_tokenizer", None)
        if bt is None or not hasattr(bt, "to_str"):
            return None
        try:
            hit = cls._BY_TOKENIZER.get(tokenizer)
        except TypeError:
            return None
        if hit is not None and (hit is False or hit.signature == cls._signature(tokenizer)):
            return hit or None
        twin = None
        try:
            cfg = json.loads(bt.to_str())
            why = _clip_pipeline(cfg)
            if why is None and getattr(tokenizer, "padding_side", "right") == "right" \
                    and getattr(tokenizer, "truncation_side", "right") == "right" and tokenizer.pad_token_id is not None \
                    and 2 <= int(tokenizer.model_max_length) <= 4096:
                twin = cls(tokenizer, cfg)
                if not twin._agrees(tokenizer, cfg):
                    twin = None
        except Exception:
            twin = None
        cls._BY_TOKENIZER[tokenizer] = twin if twin is not None else False
        return twin

    @classmethod
    def disable(cls, tokenizer) -> None:
        try:
            cls._BY_TOKENIZER[tokenizer] = False
        except TypeError:
            pass

    def _agrees(self, tokenizer, cfg) -> bool:
        vocab = cfg["model"]["vocab"]
        suffix = cfg["model"]["end_of_word_suffix"]
        words = [t[:-len(suffix)] for t in list(vocab)[-400:] if t.endswith(suffix) and t[:-len(suffix)].isascii()]
        probes = list(_PROBE_WORDS) + [" ".join(words[i:i + 7]) for i in range(0, len(words), 7)]
        probes += [f"{a}'{b}" for a, b in zip(words[::5], words[1::5])][:40]
        want = tokenizer(probes, padding=True, truncation=True)
        got = self.tokenize(tokenizer, probes)
        return all(np.array_equal(np.asarray(want[k], dtype=np.int64), got[k]) for k in ("input_ids", "attention_mask")) \
            and set(want.keys()) == {"input_ids", "attention_mask"}


def find_token_ranges(ids: np.ndarray, piece_ns: bytes, piece_off: np.ndarray, piece_len: np.ndarray, subjects,
                      forbid: str = "", subject_idx: Optional[np.ndarray] = None, normalize: bool = False):
    """``emcid_find_token_ranges[_idx]``: (first, last, status) arrays for the rows of ``ids`` (B, S) int64; with
    ``subject_idx`` row i searches ``subjects[subject_idx[i]]``, else ``subjects[i]``.  ``subjects``: strings, or what
    ``pack_strings`` made of them.  ``normalize``: the subjects are raw (the library lower-cases and strips spaces)."""
    lib = load()
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    B, S = ids.shape
    sb, soff = subjects if isinstance(subjects, tuple) else pack_strings(subjects)
    n_subjects = len(soff) - 1
    first = np.empty(B, dtype=np.int32)
    last = np.empty(B, dtype=np.int32)
    status = np.empty(B, dtype=np.uint8)
    if subject_idx is not None:
        subject_idx = np.ascontiguousarray(subject_idx, dtype=np.int32)
        if subject_idx.shape != (B,):
            raise ValueError("subject_idx must have one entry per row")
    elif n_subjects != B:
        raise ValueError("one subject per row")
    rc = lib.emcid_find_token_ranges_idx(_ptr(ids), B, S, piece_ns, _ptr(piece_off), _ptr(piece_len), len(piece_len), sb,
                                         _ptr(soff), _ptr(subject_idx) if subject_idx is not None else None, n_subjects,
                                         int(bool(normalize)), forbid.encode(), _ptr(first), _ptr(last), _ptr(status))
    if rc < 0:
        raise RuntimeError((lib.emcid_host_last_error() or b"").decode())
    return first, last, status


def build_trie_packed(ids: np.ndarray, lookup: np.ndarray, bucket: int, alloc):
    """``emcid_trie_build`` + ``emcid_trie_export``: ``alloc(nbytes)`` returns (object, address) of a host buffer (pinned, for an
    asynchronous upload); returns (object, dict(U, n_real, dmax, R_pad, n)).  Layout: include/emcid_host.h."""
    lib = load()
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    lookup = np.ascontiguousarray(lookup, dtype=np.int64)
    n, S = ids.shape
    h = lib.emcid_trie_build(_ptr(ids), n, S, _ptr(lookup), int(bucket))
    if not h:
        raise RuntimeError((lib.emcid_host_last_error() or b"").decode())
    try:
        sz = (ctypes.c_int64 * 4)()
        base = ctypes.addressof(sz)
        lib.emcid_trie_sizes(h, base, base + 8, base + 16, base + 24)
        nbytes = lib.emcid_trie_packed_bytes(h)
        buf, addr = alloc(nbytes)
        if lib.emcid_trie_export(h, addr, nbytes) != 0:
            raise RuntimeError((lib.emcid_host_last_error() or b"").decode())
    finally:
        lib.emcid_trie_destroy(h)
    return buf, dict(U=int(sz[0]), n_real=int(sz[1]), dmax=int(sz[2]), R_pad=int(sz[3]), n=n)
