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
from typing import Dict, List, Sequence, Tuple


def decode_tokens(tokenizer, token_array):
    if hasattr(token_array, "shape") and len(token_array.shape) > 1:
        return [decode_tokens(tokenizer, row) for row in token_array]
    return [tokenizer.decode([t]) for t in token_array]


class TokenRangeFinder:
    def __init__(self, tokenizer):
        self.tokenizer = tokenizer
        self._piece: Dict[int, str] = {}

    def _pieces(self, ids: Sequence[int]) -> List[str]:
        out = []
        for t in ids:
            s = self._piece.get(t)
            if s is None:
                s = self._piece[t] = self.tokenizer.decode([t])
            out.append(s)
        return out

    def batch(self, token_arrays, substrings) -> List[Tuple[int, int]]:
        """All prompts of a request list at once: ONE ``batch_decode`` for the whole-prompt strings (the same text
        ``decode`` gives per prompt) instead of one tokenizer call per prompt."""
        rows = [[int(t) for t in ids] for ids in token_arrays]
        wholes = self.tokenizer.batch_decode(rows)
        return [self(ids, sub, whole) for ids, sub, whole in zip(rows, substrings, wholes)]

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
