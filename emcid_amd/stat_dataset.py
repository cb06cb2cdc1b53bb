"""Caption dataset and length-bucketed collation for the Stage-0 statistics pass.

Host-side counterpart of the reference's dsets/stat_dataset.py (``TokenizedDataset`` :71-110,
``length_collation`` :122-150, ``make_padded_batch`` :153-163, ``flatten_masked_batch`` :166-172,
``dict_to_`` :113-119).  No network: a missing caption file is an error, not a download.
"""
import json
import os
from typing import Dict, List

import torch
from torch.nn.utils.rnn import pad_sequence
from torch.utils.data import Dataset


class TokenizedDataset(Dataset):
    """JSON list of ``{"caption": str}`` -> token ids + position ids + all-ones attention mask."""

    def __init__(self, data_path, tokenizer=None, maxlen=None):
        if not os.path.exists(data_path):
            raise FileNotFoundError(f"{data_path}: caption file missing (no network in this build; place it there)")
        with open(data_path, "r") as f:
            self.data = [row["caption"] for row in json.load(f)]
        self.tokenizer = tokenizer
        self.maxlen = maxlen

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx) -> Dict[str, torch.Tensor]:
        ids = self.tokenizer.encode(self.data[idx], truncation=True, max_length=self.maxlen)
        n = len(ids)
        return {"input_ids": torch.tensor(ids), "position_ids": torch.arange(n),
                "attention_mask": torch.ones(n, dtype=torch.long)}


def dict_to_(data, device):
    for k in data:
        data[k] = data[k].to(device, non_blocking=True)
    return data


def make_padded_batch(items: List[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    longest = max(len(it["input_ids"]) for it in items)
    if longest == 0:
        return {k: torch.zeros((0, 0), dtype=torch.long) for k in items[0]}
    live = [it for it in items if len(it["input_ids"])]
    return {k: pad_sequence([it[k] for it in live], batch_first=True) for k in items[0]}


def length_collation(token_size: int):
    """Longest-first; a sub-batch is closed when (its width) x (rows + 1) would exceed ``token_size``."""

    def collate_fn(items):
        ordered = sorted(items, key=lambda it: -len(it["input_ids"]))
        out, cur, width = [], [], 0
        for it in ordered:
            w = len(it["input_ids"])
            if w == 0:
                break
            if width * (len(cur) + 1) > token_size:
                out.append(make_padded_batch(cur))
                cur, width = [], 0
            if not cur:
                width = w
            cur.append(it)
        if cur:
            out.append(make_padded_batch(cur))
        return out

    return collate_fn


def flatten_masked_batch(data: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """(B, S, c) -> (T_attended, c): drops padded positions, keeps BOS/EOS."""
    flat = data.reshape(-1, data.size(-1))
    return flat[mask.reshape(-1).nonzero()[:, 0]]


def collate_token_lists(rows, token_size: int):
    """``length_collation(token_size)`` applied to plain token-id lists, vectorised: same ordering (longest first,
    stable), same split rule, same zero padding, plus the flat indices of the attended positions (what
    ``flatten_masked_batch`` selects) computed on the host so the device never has to run ``nonzero``.
    Returns a list of dicts {input_ids, position_ids, attention_mask (rows, width), attended (T,)} of CPU tensors."""
    import numpy as np

    order = sorted(range(len(rows)), key=lambda i: -len(rows[i]))
    groups, cur, width = [], [], 0
    for i in order:
        w = len(rows[i])
        if w == 0:
            break
        if width * (len(cur) + 1) > token_size:
            groups.append(cur)
            cur, width = [], 0
        if not cur:
            width = w
        cur.append(i)
    if cur:
        groups.append(cur)
    out = []
    for g in groups:
        lens = np.fromiter((len(rows[i]) for i in g), dtype=np.int64, count=len(g))
        w = int(lens.max())
        mask = np.arange(w)[None, :] < lens[:, None]
        ids = np.zeros((len(g), w), dtype=np.int64)
        ids[mask] = np.concatenate([np.asarray(rows[i], dtype=np.int64) for i in g])
        pos = np.where(mask, np.arange(w)[None, :], 0)
        out.append({"input_ids": torch.from_numpy(ids), "position_ids": torch.from_numpy(pos),
                    "attention_mask": torch.from_numpy(mask.astype(np.int64)),
                    "attended": torch.from_numpy(np.flatnonzero(mask.reshape(-1)))})
    return out
