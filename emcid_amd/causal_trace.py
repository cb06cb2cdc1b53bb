"""Subject-token location inside a tokenized prompt.

Host-side counterpart of the reference's experiments/causal_trace.py:1046-1103 (``decode_tokens``,
``find_token_range``).  The index it yields drives the K/Z gather, so it reproduces the reference's
character-offset walk exactly, including its special cases (``[CLS]``, ``[EOS]``/empty subject, the
right-single-quote substitution and the two-token ``ń``); pinned by tests/golden/token_ranges.json.

``TokenRangeFinder`` adds what a 3 000-prompt batch needs and the reference lacks: the per-token decode
strings are memoised per tokenizer, so the walk costs one dict lookup per token instead of one
``tokenizer.decode`` call.
"""
import unicodedata
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import host_text


def decode_tokens(tokenizer, token_array):
    if hasattr(token_array, "shape") and len(token_array.shape) > 1:
        return [decode_tokens(tokenizer, row) for row in token_array]
    return [tokenizer.decode([t]) for t in token_array]


class TokenRangeFinder:
    def __init__(self, tokenizer):
        self.tokenizer = tokenizer
        self._piece: Dict[int, str] = {}
        self._plen: Optional[np.ndarray] = None      # len(decode([t])) per token id, -1 = not decoded yet
        self._piece_ns: List[Optional[str]] = []     # decode([t]) without spaces ("\ufffd" if it holds a replacement char)
        self._suffix = False                         # see _compositional (False = not probed yet)
        self._native = None                          # (piece bytes per id, packed blob, offsets, lengths) for libemcid_host

    def _pieces(self, ids: Sequence[int]) -> List[str]:
        out = []
        for t in ids:
            s = self._piece.get(t)
            if s is None:
                s = self._piece[t] = self.tokenizer.decode([t])
            out.append(s)
        return out

    def _decode_whole(self):
        """Whole-prompt decoder: ``tokenizer.decode(ids)`` — for a tokenizers-backed HF tokenizer the backend call that
        ``decode`` wraps (transformers ``TokenizersBackend._decode``: backend decode, then ``clean_up_tokenization``, which
        only ever deletes SPACES; the walk below strips every space first, so the two are interchangeable here)."""
        bt = getattr(self.tokenizer, "_tokenizer", None)
        if bt is not None and type(self.tokenizer).__mro__[1].__name__ == "TokenizersBackend" and hasattr(bt, "decode"):
            dec = bt.decode
            return lambda ids: dec(ids, skip_special_tokens=False)
        return self.tokenizer.decode

    def _compositional(self) -> Optional[str]:
        """The end-of-word suffix if this is a byte-level BPE tokenizer, whose ``decode`` is compositional: the text is
        the UTF-8 decoding of the concatenated token bytes (tokenizers' ByteLevel decoder), then — CLIP — every
        end-of-word suffix becomes a space.  Whenever each token's own bytes are valid UTF-8 (its single-token decode has
        no U+FFFD), decode(ids) equals the concatenation of the single-token decodes up to SPACES, which the search
        strips anyway.  ``batch`` then never decodes a whole row; rows with a U+FFFD piece, or whose concatenation still
        shows the suffix (formed across a token boundary), take the whole-row decode.  None: not such a tokenizer."""
        if self._suffix is not False:
            return self._suffix
        self._suffix = None
        bt = getattr(self.tokenizer, "_tokenizer", None)
        try:
            if bt is not None and type(bt.model).__name__ == "BPE" and type(bt.decoder).__name__ == "ByteLevel":
                self._suffix = bt.model.end_of_word_suffix or ""
        except Exception:
            self._suffix = None
        return self._suffix

    def _row_strings(self, ids: np.ndarray, rows, need) -> List[Optional[str]]:
        """Space-free decoded string of every row in ``need`` (a boolean list), None elsewhere."""
        suffix = self._compositional()
        dec = self._decode_whole()
        if suffix is None:
            return [dec(r).replace(" ", "") if n else None for r, n in zip(rows, need)]
        top = int(ids.max()) + 1
        while len(self._piece_ns) < top:
            self._piece_ns.append(None)
        table = self._piece_ns
        for t in np.unique(ids).tolist():
            if table[t] is None:
                piece = self._pieces([t])[0]
                table[t] = "\ufffd" if "\ufffd" in piece else piece.replace(" ", "")
        out = []
        get = table.__getitem__
        for r, n in zip(rows, need):
            if not n:
                out.append(None)
                continue
            cat = "".join(map(get, r))
            if "\ufffd" in cat or (suffix and suffix in cat):
                cat = dec(r).replace(" ", "")
            out.append(cat)
        return out

    def _piece_lengths(self, ids: np.ndarray) -> np.ndarray:
        """len(decode([t])) for every id of the (B, S) array, from a lazily filled per-token table."""
        top = int(ids.max()) + 1
        if self._plen is None or self._plen.size < top:
            grown = np.full(top, -1, dtype=np.int64)
            if self._plen is not None:
                grown[:self._plen.size] = self._plen
            self._plen = grown
        missing = np.unique(ids[self._plen[ids] < 0])
        for t in missing.tolist():
            self._plen[t] = len(self._pieces([t])[0])
        return self._plen[ids]

    def batch(self, token_arrays, substrings) -> List[Tuple[int, int]]:
        """All prompts of a request list at once (rows of equal length, as the padded tokenizer output): the whole-prompt
        strings come from one decode per row, the character-offset walk is a cumulative sum over a per-token length
        table for the whole batch.  Rows that touch a special case of the scalar walk (``[CLS]``, ``[EOS]``/empty subject,
        the two-token n-acute, a subject that is never covered) go through ``__call__``, so every result is the scalar one."""
        try:
            ids = np.asarray(token_arrays, dtype=np.int64)
        except ValueError:          # ragged rows
            ids = np.zeros(0, dtype=np.int64)
        if ids.ndim != 2 or ids.size == 0:
            return [self(row, sub) for row, sub in zip(token_arrays, substrings)]
        B, S = ids.shape
        if self._compositional() is not None and host_text.available() and int(ids.min()) >= 0:
            return self._batch_native(ids, substrings)
        rows = ids.tolist()
        at = np.zeros(B, dtype=np.int64)
        stop = np.zeros(B, dtype=np.int64)
        scalar = []
        special = ("[CLS]", "[EOS]", "", " ")
        wholes = self._row_strings(ids, rows, [sub0 not in special for sub0 in substrings])
        for i, sub0 in enumerate(substrings):
            if sub0 in special:
                scalar.append(i)
                continue
            sub = sub0.replace(" ", "").lower()
            whole = wholes[i]
            if "’" in sub:
                whole = whole.replace("'", "’")
            if not whole.isascii():
                whole = unicodedata.normalize("NFKC", whole)
            if not sub.isascii():
                sub = unicodedata.normalize("NFKC", sub)
                if "ń" in sub:
                    scalar.append(i)
                    continue
            a = whole.find(sub)
            if a < 0:
                raise ValueError(f"subject {sub0!r} not found in tokens: {whole!r}")
            at[i], stop[i] = a, a + len(sub)
            if not sub:
                scalar.append(i)
        cum = np.cumsum(self._piece_lengths(ids), axis=1)
        started = cum > at[:, None]
        covered = cum >= stop[:, None]
        first, last = started.argmax(axis=1), covered.argmax(axis=1) + 1
        ok = started[np.arange(B), first] & covered[np.arange(B), last - 1]
        out = list(zip(first.tolist(), last.tolist()))
        for i in set(scalar) | set(np.nonzero(~ok)[0].tolist()):
            out[i] = self(rows[i], substrings[i])
        return out

    def _native_tables(self, ids: np.ndarray):
        """Per-token tables for ``emcid_find_token_ranges``, grown when the batch shows token ids not decoded before."""
        top = int(ids.max()) + 1
        if self._native is not None and self._native[4].size >= top and self._native[1] is not None \
                and self._native[4][ids.ravel()].all():
            return self._native[1], self._native[2], self._native[3]          # every token of the batch is in the tables already
        if self._native is None:
            self._native = [[], b"", np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=bool)]
        pieces, blob, off, plen, known = self._native
        if known.size < top:
            pieces.extend([b""] * (top - len(pieces)))
            plen = np.concatenate([plen, np.full(top - plen.size, -1, dtype=np.int32)])
            known = np.concatenate([known, np.zeros(top - known.size, dtype=bool)])
            blob = None
        present = np.zeros(known.size, dtype=bool)
        present[ids.ravel()] = True
        new = np.nonzero(present & ~known)[0]
        if new.size:
            for t, piece in zip(new.tolist(), self._pieces(new.tolist())):
                ns = piece.replace(" ", "")
                if ns.isascii():
                    pieces[t], plen[t] = ns.encode("ascii"), len(piece)
            known[new] = True
            blob = None
        if blob is None:
            off = np.zeros(len(pieces) + 1, dtype=np.int64)
            np.cumsum(np.fromiter(map(len, pieces), dtype=np.int64, count=len(pieces)), out=off[1:])
            blob = b"".join(pieces)
            self._native = [pieces, blob, off, plen, known]
        return blob, off, plen

    def _batch_native(self, ids: np.ndarray, substrings, index: Optional[np.ndarray] = None) -> List[Tuple[int, int]]:
        """``batch`` through libemcid_host (``emcid_find_token_ranges``): the same walk over the same per-token pieces; the
        rows it hands back (special subjects, non-ASCII, a subject it does not find) take the scalar path, errors included.
        ``index``: row i searches ``substrings[index[i]]`` (the subjects of a mass edit repeat once per template)."""
        blob, off, plen = self._native_tables(ids)
        special = ("[CLS]", "[EOS]", "", " ")
        subs = []
        for sub0 in substrings:
            sub = "" if sub0 in special else sub0.replace(" ", "").lower()
            subs.append(sub if sub.isascii() else "")
        first, last, status = host_text.find_token_ranges(ids, blob, off, plen, subs, forbid=self._compositional() or "",
                                                          subject_idx=index)
        out = list(zip(first.tolist(), last.tolist()))
        todo = np.nonzero(status)[0]
        if todo.size:
            for i in todo.tolist():
                out[i] = self(ids[i].tolist(), substrings[i if index is None else int(index[i])])
        return out

    def last_tokens(self, ids: np.ndarray, subjects: Sequence[str], index: np.ndarray, packed=None) -> np.ndarray:
        """Lookup position (``find_token_range(...)[1] - 1``, compute_z.py:2287-2290) of every row of ``ids`` (B, S) for the
        subject ``subjects[index[i]]``, as an int64 array.  ``packed``: ``host_text.pack_strings(subjects)`` if the caller
        has it already.  The native walk lower-cases and strips the subjects itself; special subjects, non-ASCII ones and
        rows it cannot serve take the scalar walk (its errors included)."""
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        index = np.asarray(index)
        if self._compositional() is not None and host_text.available() and ids.size and int(ids.min()) >= 0:
            blob, off, plen = self._native_tables(ids)
            if packed is None:
                packed = host_text.pack_strings(subjects)
            _, last, status = host_text.find_token_ranges(ids, blob, off, plen, packed, forbid=self._compositional() or "",
                                                          subject_idx=index, normalize=True)
            out = last.astype(np.int64) - 1
            for i in np.nonzero(status)[0].tolist():
                out[i] = self(ids[i].tolist(), subjects[int(index[i])])[1] - 1
            return out
        rng = self.batch(ids, [subjects[int(k)] for k in index])
        return np.fromiter((r[1] - 1 for r in rng), dtype=np.int64, count=len(rng))

    def __call__(self, token_array, substring_orig: str, whole_decoded: str = None) -> Tuple[int, int]:
        ids = [int(t) for t in token_array]
        n = len(ids)
        sub = substring_orig
        if sub == "[CLS]":
            return (0, 1)
        if sub == "[EOS]" or sub == "" or sub == " ":
            return (n - 1, n)
        sub = sub.replace(" ", "").lower()
        whole = (self.tokenizer.decode(ids) if whole_decoded is None else whole_decoded).replace(" ", "")
        if "’" in sub:
            whole = whole.replace("'", "’")
        whole = unicodedata.normalize("NFKC", whole)
        sub = unicodedata.normalize("NFKC", sub)
        at = whole.find(sub)
        if at < 0:
            raise ValueError(f"subject {substring_orig!r} not found in tokens: {whole!r}")
        stop_at = at + len(sub)
        skip_n_acute = "ń" in sub
        seen = 0
        first = last = None
        for i, piece in enumerate(self._pieces(ids)):
            if not (skip_n_acute and ids[i] == 78):
                seen += len(piece)
            if first is None and seen > at:
                first = i
            if last is None and seen >= stop_at:
                last = i + 1
                break
        return (first, last)


def find_token_range(tokenizer, token_array, substring_orig: str) -> Tuple[int, int]:
    """Returned range is [start, end); lookup index of the edit is ``end - 1`` (compute_z.py:2287-2290)."""
    return TokenRangeFinder(tokenizer)(token_array, substring_orig)
