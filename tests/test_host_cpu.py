"""CPU tests: host logic of the drop-in boundary, cache formats, and that the C-ABI library loads and
exports every symbol include/emcid_hip.h declares (no compute calls without a GPU)."""
import ctypes
import json
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import GOLDEN, REPO
from emcid_amd import hip, synthetic as syn
from emcid_amd import runningstats as rs
from emcid_amd.causal_trace import TokenRangeFinder, find_token_range
from emcid_amd.compute_z import build_prompt_batch, expand_request_prompts
from emcid_amd.emcid_hparams import EMCIDHyperParams, EMCIDXLHyperParams
from emcid_amd.nethook import StopForward, Trace, TraceDict, get_module, get_parameter
from emcid_amd.stat_dataset import TokenizedDataset, flatten_masked_batch, length_collation
from oracle import emcid_oracle as orc


def test_library_exports_every_declared_symbol():
    header = (REPO / "include" / "emcid_hip.h").read_text()
    declared = set(re.findall(r"\b(emcid_[a-z0-9_]+)\s*\(", header))
    assert declared == set(hip.EXPORTS), declared ^ set(hip.EXPORTS)
    lib = ctypes.CDLL(str(hip.lib_path()))
    for name in declared:
        assert hasattr(lib, name), name
    assert hip.load().emcid_abi_version() == hip.ABI_VERSION


def test_c_abi_rejects_bad_arguments_without_gpu():
    lib = hip.load()
    assert lib.emcid_edit_workspace_bytes(0, 3072, 768) == 0
    assert lib.emcid_edit_workspace_bytes(1000, 3072, 768) == 8 * (2 * 3072 * 3072 + 6 * (512 * 512 + 256 * 256) + 2 * 1024 * 3072 + 1024 * 768)
    rc = lib.emcid_gram_accumulate_f32(None, 10, 128, 128, None, 128, 0, None)
    assert rc == -1 and b"bad argument" in lib.emcid_last_error()
    rc = lib.emcid_cholesky_f64(None, None, 100, 100, None, None, None)
    assert rc == -1


def test_product_refuses_cpu_tensors():
    with pytest.raises(hip.EmcidHipError, match="HBM"):
        hip.gram_accumulate_(torch.zeros(8, 8), torch.zeros(4, 8))
    with pytest.raises(hip.EmcidHipError, match="HBM"):
        rs.SecondMoment().add(torch.zeros(4, 8))


def test_token_ranges_golden_and_memo():
    tok = syn.build_tokenizer()
    finder = TokenRangeFinder(tok)
    for r in json.load(open(GOLDEN / "token_ranges.json")):
        for fn in (finder, lambda ids, s: find_token_range(tok, torch.tensor(ids), s)):
            if r["range"] == "ValueError":
                with pytest.raises(ValueError):
                    fn(r["ids"], r["subject"])
            else:
                assert list(fn(r["ids"], r["subject"])) == r["range"]


def test_token_range_batch_equals_scalar_walk(monkeypatch):
    """The vectorised batch walk (cumulative piece lengths) returns what the scalar walk returns, row for row, including
    the rows it must hand to the scalar path ([CLS], [EOS], '', ' ', all-space subjects) and a missing subject."""
    tok = syn.build_tokenizer(*syn.synthetic_vocab(syllables=True))
    names = syn.syllable_names(40) + ["Vincent van Gogh", "c0042", "by", "[CLS]", "[EOS]", "", " ", "o'keeffe", "painting"]
    prompts = [t.format(n) for n in names for t in syn.ARTIST_TEMPLATES]
    subjects = [n for n in names for _ in syn.ARTIST_TEMPLATES]
    ids = tok(prompts, padding=True, truncation=True)["input_ids"]
    scalar = TokenRangeFinder(tok)
    want = [scalar(r, s) for r, s in zip(ids, subjects)]
    orc_want = [orc.find_token_range(tok, torch.tensor(r), s) for r, s in zip(ids, subjects)]
    assert [tuple(w) for w in want] == [tuple(w) for w in orc_want]
    from emcid_amd import host_text
    for native in (True, False):        # libemcid_host's walk, and the numpy walk it replaces
        monkeypatch.setattr(host_text, "available", lambda native=native: native)
        got = TokenRangeFinder(tok).batch(ids, subjects)
        assert [tuple(g) for g in got] == [tuple(w) for w in want]
        with pytest.raises(ValueError):
            TokenRangeFinder(tok).batch(ids[:3], ["zzzz", subjects[1], subjects[2]])
        # a row with a piece that is not valid UTF-8 on its own (U+FFFD) takes the whole-row decode: same answer
        poisoned = TokenRangeFinder(tok)
        poisoned.batch(ids[:4], subjects[:4])
        if native:
            assert poisoned._native is not None
            poisoned._native[3][ids[0][3]] = -1
        else:
            assert poisoned._native is None
            poisoned._piece_ns[ids[0][3]] = "\ufffd"
        assert [tuple(g) for g in poisoned.batch(ids, subjects)] == [tuple(w) for w in want]
        # ragged rows (no common length): the scalar path row by row
        ragged = [r[:len(r) - (i % 2)] for i, r in enumerate(ids[:6])]
        assert TokenRangeFinder(tok).batch(ragged, subjects[:6]) == [scalar(r, s) for r, s in zip(ragged, subjects[:6])]
    assert TokenRangeFinder(tok)._compositional() == "</w>"


def test_native_token_ranges_hand_back_what_they_cannot_serve():
    """The suffix formed across tokens ('<', '/', 'w', '>'), non-ASCII subjects and pieces, and subjects that appear only
    partially: the native walk flags them and the scalar walk answers, so the batch equals the scalar walk everywhere."""
    tok, _ = _rich_pair(0)
    prompts = ["a photo of </w> thing", "a photo of café", "the thing and the thing", "an image of the-thing!", "thing"]
    subjects = ["thing", "café", "thing", "the-thing", "thing"]
    ids = np.asarray(tok(prompts, padding=True, truncation=True)["input_ids"])
    f = TokenRangeFinder(tok)
    want = []
    for r, s_ in zip(ids.tolist(), subjects):
        try:
            want.append(tuple(TokenRangeFinder(tok)(r, s_)))
        except ValueError:
            want.append("ValueError")
    assert want[1] == "ValueError" and want.count("ValueError") == 1     # 'é' is outside this vocabulary: the reference raises too
    keep = [0, 2, 3, 4]
    assert [tuple(g) for g in f.batch(ids[keep], [subjects[i] for i in keep])] == [want[i] for i in keep]
    assert f._native is not None
    with pytest.raises(ValueError):
        f.batch(ids, subjects)


def test_fast_tokenize_equals_public_call():
    from emcid_amd.compute_z import tokenize_lists
    tok = syn.build_tokenizer(*syn.synthetic_vocab(syllables=True))
    prompts = [t.format(n) for n in syn.syllable_names(30) + ["a " * 90 + "b"] for t in syn.ARTIST_TEMPLATES]   # one over 77 tokens
    pub = tok(prompts, padding=True, truncation=True)
    got = tokenize_lists(tok, prompts)
    assert got["input_ids"].tolist() == pub["input_ids"] and got["attention_mask"].tolist() == pub["attention_mask"]
    assert got["input_ids"].shape[1] == 77
    few = tokenize_lists(tok, prompts[:3])          # small batches take the public call
    assert few["input_ids"].tolist() == tok(prompts[:3], padding=True, truncation=True)["input_ids"]


def test_vectorised_trie_is_the_prefix_set_up_to_the_lookup_tokens():
    from emcid_amd import clip_forward as cf
    rng = np.random.default_rng(5)
    B, S = 200, 9
    ids = rng.integers(0, 3, size=(B, S))
    ids[:, 0] = 7
    lookup = rng.integers(0, S, size=B)
    trie = cf.build_trie(ids, lookup.tolist(), "cpu", bucket=16)
    tok_np, anc_np, dep = trie.token.numpy(), trie.anc.numpy(), trie.depth.numpy()
    want = {tuple(ids[b, :p + 1]) for b in range(B) for p in range(lookup[b] + 1)}
    got = {tuple(int(tok_np[a]) for a in anc_np[u, :dep[u] + 1]) for u in range(trie.n_nodes)}
    assert got == want and trie.n_nodes == len(want)
    for b in range(B):          # every prompt's lookup node spells the prompt up to its lookup token
        u = int(trie.lookup_node[b])
        assert dep[u] == lookup[b] and [int(tok_np[a]) for a in anc_np[u, :dep[u] + 1]] == ids[b, :lookup[b] + 1].tolist()
        assert int(trie.query_rows[int(trie.lookup_in_query[b])]) == u
    U = trie.token.shape[0]
    assert U % 16 == 0 and trie.query_rows.numel() % 16 == 0
    assert (anc_np[trie.n_nodes:, 0] == np.arange(trie.n_nodes, U)).all() and (dep[trie.n_nodes:] == 0).all()
    for u in range(trie.n_nodes):       # parents precede children (the forward relies on nothing else about the order)
        assert (anc_np[u, :dep[u]] < u).all() and anc_np[u, dep[u]] == u


@pytest.mark.parametrize("B,S,vocab,bucket,seed", [(200, 9, 3, 16, 5), (3000, 7, 50, 256, 6), (17, 1, 4, 1, 7), (64, 77, 2, 256, 8)])
def test_native_trie_equals_numpy_trie(B, S, vocab, bucket, seed):
    """libemcid_host's emcid_trie_build (what the edit path runs) and the numpy construction give the same arrays: node
    numbering, ancestor chains, padding, query rows."""
    from emcid_amd import clip_forward as cf, host_text
    if not host_text.available():
        pytest.skip("libemcid_host.so not built")
    rng = np.random.default_rng(seed)
    ids = rng.integers(0, vocab, size=(B, S)).astype(np.int64)
    ids[:, 0] = 7
    lookup = rng.integers(0, S, size=B).astype(np.int64)
    a = cf.build_trie(ids, lookup, "cpu", bucket=bucket)
    b = cf.build_trie_numpy(ids, lookup, "cpu", bucket=bucket)
    assert a.n_nodes == b.n_nodes and a.n_tokens_dense == b.n_tokens_dense
    for f in ("token", "depth", "anc", "lookup_node", "query_rows", "lookup_in_query"):
        x, y = getattr(a, f), getattr(b, f)
        assert x.dtype == y.dtype and x.shape == y.shape and torch.equal(x, y), f
    with pytest.raises(RuntimeError):
        host_text.build_trie_packed(ids, np.full(B, S, dtype=np.int64), bucket, lambda n: (None, 0))      # lookup outside the row
    t = cf.build_trie(ids, lookup, "cpu", bucket=bucket, tail=np.arange(B + 1) * 3)       # the caller's array behind the image
    assert t.tail.dtype == torch.int64 and t.tail.tolist() == (np.arange(B + 1) * 3).tolist() and torch.equal(t.anc, b.anc)


def test_prompt_batch_matches_oracle_lookup():
    tok = syn.build_tokenizer()
    reqs = syn.make_requests(9, ragged=True)
    reqs[3]["source"] = "Vincent van Gogh"
    b = build_prompt_batch(tok, reqs, "cpu", truncate=False)
    prompts, subjects, counts = orc.expand_requests(reqs)
    enc = orc.tokenize_prompts(prompts, tok, "cpu")
    look = [orc.find_token_range(tok, ids, w)[-1] - 1 for ids, w in zip(enc["input_ids"], subjects)]
    assert b.lookup.tolist() == look and torch.equal(b.inputs["input_ids"], enc["input_ids"])
    assert b.seg.tolist() == np.cumsum([0] + counts).tolist() and b.n_requests == 9
    # default: columns after the last lookup token are dropped (causal encoder: they cannot matter)
    bt = build_prompt_batch(tok, reqs, "cpu")
    keep = max(look) + 1
    assert bt.inputs["input_ids"].shape[1] == keep < enc["input_ids"].shape[1]
    assert torch.equal(bt.inputs["input_ids"], enc["input_ids"][:, :keep]) and bt.lookup.tolist() == look
    assert torch.equal(bt.inputs["attention_mask"], enc["attention_mask"][:, :keep])


def test_expand_requests_source_prompts_branch():
    reqs = [{"source": "a", "dest": "b", "prompts": ["x {}", "y {}"], "source_prompts": ["pre a one", "pre a two"]},
            {"source": "c", "dest": "b", "prompts": ["x {}", "y {}"], "source_prompts": ["pre c one", "pre c two"]}]
    p, s, c = expand_request_prompts(reqs)
    assert p == ["pre a one", "pre a two", "pre c one", "pre c two"] and s == ["a", "a", "c", "c"] and c == [2, 2]
    assert (p, s, c) == orc.expand_requests(reqs)


def test_nethook_prefix_tolerance_and_stop():
    te = syn.build_text_encoder("toy")
    m1 = get_module(te, "text_model.encoder.layers.2.mlp.fc2")
    assert m1 is get_module(te, "encoder.layers.2.mlp.fc2")
    assert get_parameter(te, "text_model.encoder.layers.2.mlp.fc2.weight") is m1.weight
    with pytest.raises(LookupError):
        get_module(te, "encoder.layers.99.mlp.fc2")
    tok = syn.build_tokenizer()
    enc = tok(["a photo of tench"], return_tensors="pt")
    ran = []
    h = get_module(te, "encoder.layers.3").register_forward_hook(lambda *a: ran.append(3))
    with torch.no_grad(), Trace(te, "encoder.layers.2.mlp.fc2", retain_input=True, stop=True) as tr:
        te(**enc)
    h.remove()
    assert ran == [] and tr.input.shape[-1] == 128 and tr.output.shape[-1] == 32
    with torch.no_grad(), TraceDict(te, ["encoder.layers.0.mlp.fc2", "encoder.layers.1.mlp.fc2"], retain_input=True) as td:
        te(**enc)
    assert td["encoder.layers.1.mlp.fc2"].input.shape[-1] == 128


def test_length_collation_matches_oracle(tmp_path):
    tok = syn.build_tokenizer()
    caps = syn.write_captions(tmp_path / "c.json", 120, seed=5)
    ds = TokenizedDataset(str(tmp_path / "c.json"), tok)
    items = [ds[i] for i in range(100)]
    got = length_collation(300)(items)
    ref = orc.length_sorted_subbatches([it["input_ids"].tolist() for it in items], 300)
    assert len(got) == len(ref)
    for g, r in zip(got, ref):
        pb = orc.pad_batch(r)
        for k in ("input_ids", "position_ids", "attention_mask"):
            assert torch.equal(g[k], pb[k])
    from emcid_amd.stat_dataset import collate_token_lists
    fast = collate_token_lists([it["input_ids"].tolist() for it in items], 300)
    assert len(fast) == len(got)
    for f, g in zip(fast, got):
        for k in ("input_ids", "position_ids", "attention_mask"):
            assert torch.equal(f[k], g[k])
        assert torch.equal(f["attended"], g["attention_mask"].reshape(-1).nonzero()[:, 0])
    data = torch.arange(2 * 3 * 4, dtype=torch.float32).reshape(2, 3, 4)
    mask = torch.tensor([[1, 1, 0], [1, 0, 0]])
    assert flatten_masked_batch(data, mask).shape == (3, 4)
    with pytest.raises(FileNotFoundError):
        TokenizedDataset(str(tmp_path / "missing.json"), tok)


def test_fixed_random_subset_and_shards():
    s = rs.FixedRandomSubsetSampler(range(1000), end=300, seed=1)
    assert list(s) == orc.fixed_random_subset(1000, 300, seed=1)
    parts = [list(s.shard(r, 3)) for r in range(3)]
    assert sum(parts, []) == list(s)                       # partitioned, not re-drawn
    assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_npz_cache_protocol_roundtrip(tmp_path):
    st = rs.CombinedStat(mom2=rs.SecondMoment())
    m = np.arange(16, dtype=np.float32).reshape(4, 4)
    st.load_state_dict({"mom2.count": 7, "mom2.mom2": m, "mom2.constructor": "x"})
    f = tmp_path / "a" / "s.npz"
    rs.save_cached_state(f, st, {"sample_size": 50})
    with np.load(f) as z:
        assert sorted(z.files) == ["mom2.constructor", "mom2.count", "mom2.mom2", "sample_size"]
        assert str(z["mom2.constructor"]) == "util.runningstats.SecondMoment()"
    assert rs.load_cached_state(f, {"sample_size": 51}, quiet=True) is None     # invalidated by sample_size
    dat = rs.load_cached_state(f, {"sample_size": 50}, quiet=True)
    st2 = rs.CombinedStat(mom2=rs.SecondMoment())
    st2.load_state_dict(dat)
    assert st2.mom2.count == 7 and torch.equal(st2.mom2.moment(), torch.from_numpy(m) / 7)
    with rs.cache_load_enabled(False):
        assert rs.load_cached_state(f, {"sample_size": 50}, quiet=True) is None
    # None <-> NaN-boxed null
    boxed = rs.box_numpy_null({"a": None, "b": 3})
    assert rs.is_null_numpy_value(boxed["a"]) and rs.unbox_numpy_null(boxed) == {"a": None, "b": 3}
    assert not rs.is_null_numpy_value(np.array(np.nan))
    # tally: cached -> empty loader
    st3 = rs.CombinedStat(mom2=rs.SecondMoment())
    assert list(rs.tally(st3, None, cache=f, sample_size=50, quiet=True)) == [] and st3.mom2.count == 7


def test_stored_npz_parser_equals_numpy_load_or_declines(tmp_path):
    """read_npz_stored: the members numpy.load gives, same dtypes, same bytes, writable (torch.from_numpy takes them without a
    warning); anything it does not parse — compressed, pickled, truncated, not a zip — is None, i.e. numpy.load's business."""
    import warnings
    g = np.random.default_rng(3)
    dat = {"mom2.count": np.array(12345), "mom2.mom2": g.standard_normal((33, 33)).astype(np.float32),
           "mom2.constructor": np.array("util.runningstats.SecondMoment()"), "sample_size": np.array(50),
           "f64": g.standard_normal(7), "i8": np.arange(5, dtype=np.int8), "empty": np.zeros((0, 4), np.float32)}
    f = tmp_path / "s.npz"
    np.savez(f, **dat)
    got = rs.read_npz_stored(f)
    with np.load(f) as z:
        assert sorted(got) == sorted(z.files)
        for k in z.files:
            assert got[k].dtype == z[k].dtype and got[k].shape == z[k].shape and np.array_equal(got[k], z[k])
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert torch.equal(torch.from_numpy(got["mom2.mom2"]), torch.from_numpy(dat["mom2.mom2"]))
    # the loader uses it
    st = rs.CombinedStat(mom2=rs.SecondMoment())
    st.load_state_dict(rs.load_cached_state(f, {"sample_size": 50}, quiet=True))
    assert st.mom2.count == 12345
    # declined cases
    np.savez_compressed(tmp_path / "c.npz", **dat)
    assert rs.read_npz_stored(tmp_path / "c.npz") is None
    np.savez(tmp_path / "o.npz", a=np.array([{"x": 1}], dtype=object))
    assert rs.read_npz_stored(tmp_path / "o.npz") is None
    raw = f.read_bytes()
    (tmp_path / "t.npz").write_bytes(raw[:len(raw) // 2])
    assert rs.read_npz_stored(tmp_path / "t.npz") is None
    (tmp_path / "n.npz").write_bytes(b"not a zip at all" * 8)
    assert rs.read_npz_stored(tmp_path / "n.npz") is None
    assert rs.read_npz_stored(tmp_path / "missing.npz") is None
    # and the loader still serves the compressed file through numpy
    assert rs.load_cached_state(tmp_path / "c.npz", {"sample_size": 50}, quiet=True)["mom2.count"] == 12345


def test_hparams_load_shipped_schema(tmp_path):
    d = syn.sd_hparams_dict()
    p = tmp_path / "h.json"
    json.dump(d, open(p, "w"))
    hp = EMCIDHyperParams.from_json(p)
    assert hp.layers == [7, 8, 9, 10] and hp.num_edit_tokens == 1 and hp.edit_weight == 0.5
    x = EMCIDXLHyperParams.from_dict(syn.sdxl_hparams_dict())
    assert x.layers_2 == [26, 27, 28, 29, 30] and x.mom2_update_weight_2 == 10000
    with pytest.raises(TypeError):
        EMCIDHyperParams(**{**d, "unknown_field": 1})


def test_vstar_paths_and_loading(tmp_path):
    from emcid_amd import emcid_main as em
    hp = EMCIDHyperParams(**syn.sd_hparams_dict())
    r = {"source": "tench", "dest": "goldfish", "source_cat": "fish"}
    assert em.vstar_cache_file("c/", r, hp, 3).name == "source_tench_dest_goldfish.npz"
    assert em.vstar_cache_file("c/", r, hp, 3, "_2").name == "source_tench_dest_goldfish_2.npz"
    hp.objective = "esd"
    assert em.vstar_cache_file("c/", r, hp, 3).name == "source_tench.npz"
    hp.objective, hp.sld_supervision = "ablate-dest", True
    assert em.vstar_cache_file("c/", r, hp, 3).name == "source_fish_3.npz"
    assert em.vstar_cache_file(None, r, hp, 3) is None
    hp.sld_supervision = False
    reqs = syn.make_requests(4)
    cache = str(tmp_path / "cache") + "/"
    vs = syn.write_vstar_cache(cache, reqs, 32, seed=4)
    got = em.load_v_stars(reqs, hp, cache)
    assert got.dtype == torch.float32 and np.array_equal(got.numpy(), vs)
    assert torch.equal(got.t(), orc.load_vstars(cache, reqs))                    # zs = rows transposed
    with pytest.raises(NotImplementedError):
        em.load_v_stars(syn.make_requests(5), hp, cache)
    filled = em.load_v_stars(syn.make_requests(5), hp, cache, stage1=lambda req, sfx: torch.ones(32))
    assert filled.shape == (5, 32) and (Path(cache) / "source_c0004_dest_a realist artist.npz").exists()
    # the direct npz parser: only the file np.savez(f, v_star=...) writes; everything else goes through np.load
    f0 = str(syn.vstar_cache_path(cache, reqs[0]))
    assert np.array_equal(em._npz_single_array(open(f0, "rb").read(), "v_star"), vs[0])
    np.savez_compressed(tmp_path / "c.npz", v_star=vs[1])
    assert em._npz_single_array(open(tmp_path / "c.npz", "rb").read(), "v_star") is None
    assert np.array_equal(em._read_vstar(str(tmp_path / "c.npz")), vs[1])
    np.savez(tmp_path / "d.npz", other=vs[2], v_star=vs[3])
    assert em._npz_single_array(open(tmp_path / "d.npz", "rb").read(), "v_star") is None
    assert np.array_equal(em._read_vstar(str(tmp_path / "d.npz")), vs[3])
    # a rewritten file is re-read (the in-process copy is keyed by mtime and size)
    import os, time as _t
    np.savez(f0, v_star=vs[0] + 1)
    os.utime(f0, ns=(_t.time_ns(), _t.time_ns() + 10_000_000))
    assert np.array_equal(em.load_v_stars(reqs, hp, cache).numpy()[0], vs[0] + 1)
    (tmp_path / "e.npz").write_bytes(b"PK\x03\x04 not a zip")
    hp2 = EMCIDHyperParams(**syn.sd_hparams_dict())
    assert em._npz_single_array((tmp_path / "e.npz").read_bytes(), "v_star") is None


def test_upd_matrix_match_shape():
    from emcid_amd.emcid_main import upd_matrix_match_shape
    m = torch.zeros(3, 5)
    assert upd_matrix_match_shape(m, torch.Size([3, 5])) is m
    assert upd_matrix_match_shape(m, torch.Size([5, 3])).shape == (5, 3)
    assert upd_matrix_match_shape(torch.zeros(4, 6), torch.Size([4, 6, 1, 1])).shape == (4, 6, 1, 1)
    with pytest.raises(ValueError):
        upd_matrix_match_shape(m, torch.Size([2, 2]))


def test_export_and_load_edited_weights(tmp_path):
    from emcid_amd import emcid_main as em
    te = syn.build_text_encoder("toy")
    hp = EMCIDHyperParams(**syn.sd_hparams_dict(layers=(1, 3)))
    w = get_parameter(te, "encoder.layers.3.mlp.fc2.weight")
    with torch.no_grad():
        w += 1.0
    names = em.export_edited_weights(te, hp, tmp_path / "out" / "edit.safetensors")
    assert names == ["text_model.encoder.layers.1.mlp.fc2.weight", "text_model.encoder.layers.3.mlp.fc2.weight"]
    fresh = syn.build_text_encoder("toy")
    assert not torch.equal(get_parameter(fresh, names[1]), w)
    assert em.load_edited_weights(fresh, tmp_path / "out" / "edit.safetensors") == names
    assert torch.equal(get_parameter(fresh, names[1]), w)


def test_cross_attn_host_side(tmp_path):
    """Layer enumeration order, v* cache format and error behaviour of the cross-attention path without a GPU."""
    import numpy as np
    from conftest import load_golden
    from emcid_amd import emcid_main as em, layer_stats as ls, synthetic as syn
    from emcid_amd.emcid_hparams import EMCIDHyperParams
    from emcid_amd.hip import EmcidHipError
    z, meta = load_golden("toy_xattn")
    pipe = syn.add_unet(syn.build_pipe("toy", "cpu"), "toy")
    names = ls.get_all_cross_attn_kv_layer_names(pipe)
    assert names == meta["layer_names"] and len(names) == 32
    assert ls.get_to_edit_layername_unet("cross-v", "mid_block", 0, 0) == "mid_block.attentions.0.transformer_blocks.0.attn2.to_v"
    reqs = meta["requests"][:3]
    cache = str(tmp_path / "c") + "/"
    dims = {n: ls.get_attr_through_name(pipe.unet, n).out_features for n in names}
    vs = syn.write_xattn_vstar_cache(cache, reqs, dims, seed=3)
    got = em.load_v_stars_cross_attn(reqs, cache, names)
    for n in names:
        np.testing.assert_array_equal(got[n].numpy(), vs[n])
    with pytest.raises(NotImplementedError):
        em.load_v_stars_cross_attn(reqs + [{"source": "nobody", "prompts": ["{}"]}], cache, names)
    # statistics are served from the reference-format cache file without touching the GPU
    syn.write_stats_cache(tmp_path / "s", names[:1], 32, 1000, seed=2, t=64, model_name="unet")
    st = ls.layer_stats_cross_attn_kv(pipe, names[0], tmp_path / "s", sample_size=1000, precision="float32")
    assert st.mom2.moment().shape == (32, 32)
    # the product never falls back to the CPU
    hp = EMCIDHyperParams(**syn.sd_hparams_dict(mom2_n_samples=1000))
    with pytest.raises(EmcidHipError):
        em.apply_emcid_to_cross_attn(pipe, reqs, hp, "cpu", cache_name=cache, stats_dir=str(tmp_path / "s"), verbose=False)


def test_instruction_driver_host_side(tmp_path):
    """run_emcid.load_instruction on the reference's own instruction + hparams files (config-1 fixture): hparams class,
    set_weights overrides (emcid_test.py:924-930), cache prefix, rejection of an unknown checkpoint."""
    from conftest import load_golden
    from emcid_amd import run_emcid
    z, meta = load_golden("config1_van_gogh")
    ins, hp_file = meta["instruction"], meta["hparams_file"]
    (tmp_path / "hparams").mkdir()
    json.dump(hp_file, open(tmp_path / "hparams" / f"{ins['hparams']}.json", "w"))
    json.dump(ins, open(tmp_path / "ins.json", "w"))
    got, hp, cache = run_emcid.load_instruction(tmp_path / "ins.json", tmp_path / "hparams")
    assert isinstance(hp, EMCIDHyperParams) and not isinstance(hp, EMCIDXLHyperParams)
    assert hp.mom2_update_weight == ins["mom2_weight"] == 4000 and hp.edit_weight == ins["edit_weight"]
    assert hp_file["mom2_update_weight"] == 10000                       # the file's own value is overridden
    assert hp.layers == [7, 8, 9, 10] and cache == f"cache/{ins['hparams']}/"
    assert got["requests"][0]["source"] == "Vincent van Gogh"
    bad = dict(ins, model_ckpt="sd-v9")
    json.dump(bad, open(tmp_path / "bad.json", "w"))
    with pytest.raises(ValueError):
        run_emcid.load_instruction(tmp_path / "bad.json", tmp_path / "hparams")
    hp2 = run_emcid.set_weights(EMCIDHyperParams(**syn.sd_hparams_dict()), None, 0.7)
    assert hp2.mom2_update_weight == 4000 and hp2.edit_weight == 0.7


def _trie_prefix_multiset(trie, count=None):
    tok = trie.token.numpy()
    anc = trie.anc.numpy()
    dep = trie.depth.numpy()
    out = {}
    for u in range(trie.n_nodes):
        key = tuple(int(tok[a]) for a in anc[u, :dep[u] + 1])
        assert key not in out, "a prefix appears as two nodes"
        out[key] = 1 if count is None else int(count[u])
    return out


def test_packed_trie_is_the_prefix_multiset_of_the_captions():
    """Stage 0's numpy level-wise trie: one node per distinct prefix, count = captions through it; same node set as the
    edit path's Python trie when every last token is the lookup. Padding rows attend to themselves only."""
    from collections import Counter
    from emcid_amd import clip_forward as cf
    rng = np.random.default_rng(3)
    seqs = []
    for _ in range(300):
        n = int(rng.integers(2, 12))
        seqs.append([7] + [int(t) for t in rng.integers(0, 4, size=n - 1)])   # tiny vocabulary: many shared prefixes
    seqs.append(list(seqs[0]))                                                # an exact duplicate caption
    want = Counter(tuple(s[:i + 1]) for s in seqs for i in range(len(s)))
    trie, count, nodes = cf.build_trie_packed(seqs, "cpu", bucket=16, return_nodes=True)
    tok_np, anc_np = trie.token.numpy(), trie.anc.numpy()
    for i, sq in enumerate(seqs):          # the node of (sequence, position) spells exactly that prefix
        for pos in (0, len(sq) // 2, len(sq) - 1):
            u = nodes[i, pos]
            assert [int(tok_np[a]) for a in anc_np[u, :pos + 1]] == sq[:pos + 1]
        assert (nodes[i, len(sq):] == -1).all()
    got = _trie_prefix_multiset(trie, count.numpy())
    assert got == dict(want)
    assert int(count.sum()) == sum(len(s) for s in seqs)
    U = trie.token.shape[0]
    assert U % 16 == 0 and U >= trie.n_nodes
    anc = trie.anc.numpy()
    assert (anc[trie.n_nodes:, 0] == np.arange(trie.n_nodes, U)).all() and (trie.depth.numpy()[trie.n_nodes:] == 0).all()
    # the edit path's trie over the same sequences (padded to one length, lookup = last token) has the same prefixes
    lmax = max(len(s) for s in seqs)
    padded = [s + [0] * (lmax - len(s)) for s in seqs]
    ref = cf.build_trie(padded, [len(s) - 1 for s in seqs], "cpu", bucket=16)
    assert set(_trie_prefix_multiset(ref)) == set(got)


def test_uce_host_side():
    """Row windows (uce_train.py:109-127), the doubled projection list and its detached first entries (:232-260), the
    text formatting (:52-66) and the refusal to run on a CPU pipe."""
    from conftest import load_golden, uce_pipe_from_golden
    from emcid_amd import uce_train as uce
    z, meta = load_golden("toy_uce")
    pipe = uce_pipe_from_golden(z)
    old, new, ret = uce._format_texts(meta["old"], meta["new"], None)
    assert new[1] == " " and ret == [""] and old == meta["old"]
    texts = [t for pr in zip(old, new) for t in pr]
    ti = orc._uce_tokens(pipe.tokenizer, texts)
    S = ti.input_ids.shape[1]
    o_flat, n_flat, seg = uce.row_windows(ti.attention_mask.numpy(), len(old), S)
    r0 = 0
    for i in range(len(old)):
        (o0, o1), (n0, n1) = orc._uce_row_windows(ti.attention_mask[2 * i:2 * i + 2], S)
        m = o1 - o0
        assert m == n1 - n0
        assert (o_flat[r0:r0 + m] == 2 * i * S + np.arange(o0, o1)).all()
        assert (n_flat[r0:r0 + m] == (2 * i + 1) * S + np.arange(n0, n1)).all()
        assert (seg[r0:r0 + m] == i).all()
        r0 += m
    assert r0 == len(o_flat)
    entries, attached = uce.projection_entries(pipe, with_to_k=True)
    assert len(entries) == 64 and sum(attached) == 32 and len({e for e, a in zip(entries, attached) if a}) == 32
    assert all(e.endswith(".to_v") for e in entries[:32]) and all(e.endswith(".to_k") for e in entries[32:])
    want = meta["cases"]["ca_replace_subset"]
    e_v, a_v = uce.projection_entries(pipe, with_to_k=False)
    assert sorted(e_v[j] for j in want["layers_to_edit"] if a_v[j]) == sorted(meta["changed"]["ca_replace_subset"])
    with pytest.raises(hip.EmcidHipError):
        uce.edit_model_uce(pipe, meta["old"], meta["new"], None)
    with pytest.raises(hip.EmcidHipError):
        uce.edit_text_encoder_uce(pipe, meta["old"], meta["new"], None, layer_to_edit=2)


# ---- libemcid_host.so: native CLIP BPE + token ranges (include/emcid_host.h) ---------------------------------------------------
def _native_pair():
    from emcid_amd import host_text
    tok = syn.build_tokenizer(*syn.synthetic_vocab(syllables=True))
    twin = host_text.NativeClipBpe.for_tokenizer(tok)
    assert twin is not None, "the synthetic CLIP tokenizer must get a native twin"
    return tok, twin


def _rich_pair(seed=0, n_merges=3000):
    """A CLIP tokenizer over ALL printable ASCII with random merges (short tokens favoured, so that chains of competing
    merges of different rank occur inside ordinary words) and its native twin."""
    from emcid_amd import host_text
    rng = np.random.default_rng(seed)
    chars = [chr(c) for c in range(0x21, 0x7f)]
    vocab = {}
    for c in chars:
        vocab[c] = len(vocab)
    for c in chars:
        vocab[c + "</w>"] = len(vocab)
    open_toks, all_toks, merges, seen = list(chars), list(vocab), [], set()
    common = list("etaoinshrdlu'.,!")
    while len(merges) < n_merges:
        pool_a = open_toks if rng.random() < 0.5 else common
        a = pool_a[rng.integers(len(pool_a))]
        b = all_toks[rng.integers(len(all_toks))] if rng.random() < 0.6 else common[rng.integers(len(common))] + \
            ("</w>" if rng.random() < 0.3 else "")
        if (a, b) in seen or len(a + b) > 9 or a.endswith("</w>"):
            continue
        seen.add((a, b))
        merges.append((a, b))
        if a + b not in vocab:
            vocab[a + b] = len(vocab)
            all_toks.append(a + b)
            if not b.endswith("</w>"):
                open_toks.append(a + b)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    tok = syn.build_tokenizer(vocab, merges)
    twin = host_text.NativeClipBpe.for_tokenizer(tok)
    assert twin is not None
    return tok, twin


def test_host_library_exports_every_declared_symbol():
    import ctypes
    import re
    from emcid_amd import host_text
    header = (Path(__file__).resolve().parents[1] / "include" / "emcid_host.h").read_text()
    names = set(re.findall(r"\b(emcid_[a-z0-9_]+)\s*\(", header))
    assert {"emcid_bpe_create", "emcid_bpe_encode_batch", "emcid_find_token_ranges", "emcid_host_abi_version"} <= names
    lib = ctypes.CDLL(str(host_text.lib_path()))
    for n in names:
        assert hasattr(lib, n), n
    assert host_text.load().emcid_host_abi_version() == host_text.ABI_VERSION


def test_native_bpe_matches_hf_on_request_prompts():
    from emcid_amd.compute_z import expand_request_prompts
    tok, twin = _native_pair()
    for names in ("syllable", None):
        reqs = syn.make_requests(300, names=names) if names else syn.make_requests(64, ragged=True)
        prompts, _, _ = expand_request_prompts(reqs)
        want = tok(prompts, padding=True, truncation=True)
        got = twin.tokenize(tok, prompts)
        assert np.array_equal(np.asarray(want["input_ids"]), got["input_ids"])
        assert np.array_equal(np.asarray(want["attention_mask"]), got["attention_mask"])


def test_native_bpe_matches_hf_on_arbitrary_text():
    """Property test: printable ASCII soup (contractions, digits, punctuation runs, odd white space), long rows that get
    truncated, and rows the library must hand back (non-ASCII, control bytes, special-token syntax)."""
    from hypothesis import given, settings, strategies as st
    pairs = [_native_pair(), _rich_pair(0), _rich_pair(1, 800)]
    alphabet = st.sampled_from(list("abcdefgopqrstuvwxyzABCTZ '\t\n0123456789.,!?-_()[]'\"<|>#") + ["'s", "'re", " 'll", "n't", "'D"])
    text = st.lists(alphabet, max_size=60).map("".join)
    odd = st.sampled_from(["café au lait", "naïve", "á", "你好", "x\x00y", "x\x1fy", "a\x7fb", "<|endoftext|> hi",
                           "pre <|startoftext|>", "emoji \U0001f600", "nbsp here", "’s"])

    @settings(max_examples=150, deadline=None)
    @given(st.lists(st.one_of(text, text, text, odd), min_size=1, max_size=12), st.integers(0, 3))
    def check(rows, long_rows):
        rows = rows + ["the quick brown fox " * 30 + r for r in rows[:long_rows]]
        for tok, twin in pairs:
            want = tok(rows, padding=True, truncation=True)
            got = twin.tokenize(tok, rows)
            assert np.array_equal(np.asarray(want["input_ids"]), got["input_ids"]), rows
            assert np.array_equal(np.asarray(want["attention_mask"]), got["attention_mask"]), rows
    check()
    # and the flags themselves: served rows are not flagged, foreign rows are
    tok, twin = pairs[1]
    _, _, fb = twin.encode(["every printable: !\"#$%&'()*+,-./:;<=>?@[\\]^_`{}~ 0123456789 The End"])
    assert not fb.any()
    _, _, fb = twin.encode(["plain ascii", "café", "<|endoftext|>", "x\x00y", "tab\tok"])
    assert fb.tolist() == [False, True, True, True, False]


def test_native_bpe_rejects_a_foreign_pipeline():
    """A tokenizer whose serialized pipeline is not exactly CLIP's gets no twin (the HF path is used)."""
    import json
    from emcid_amd import host_text
    tok = syn.build_tokenizer(*syn.synthetic_vocab(syllables=True))
    cfg = json.loads(tok._tokenizer.to_str())
    assert host_text._clip_pipeline(cfg) is None
    for mutate in (lambda c: c["normalizer"]["normalizers"].pop(),
                   lambda c: c["pre_tokenizer"]["pretokenizers"][0]["pattern"].update(Regex=r"\w+"),
                   lambda c: c["model"].update(dropout=0.1),
                   lambda c: c["model"].update(end_of_word_suffix=""),
                   lambda c: c["post_processor"].update(type="TemplateProcessing"),
                   lambda c: c["added_tokens"].append({"id": 1, "content": "photo"})):
        c = json.loads(json.dumps(cfg))
        mutate(c)
        assert host_text._clip_pipeline(c) is not None


def test_tokenize_lists_uses_the_native_twin_and_agrees():
    from emcid_amd import host_text
    from emcid_amd.compute_z import expand_request_prompts, tokenize_lists
    tok, twin = _native_pair()
    prompts, _, _ = expand_request_prompts(syn.make_requests(40, names="syllable"))
    prompts[3] = "a photo of café " + prompts[3]
    calls = []
    orig = host_text.NativeClipBpe.tokenize
    try:
        host_text.NativeClipBpe.tokenize = lambda self, t, p: (calls.append(len(p)), orig(self, t, p))[1]
        got = tokenize_lists(tok, prompts)
    finally:
        host_text.NativeClipBpe.tokenize = orig
    assert calls == [len(prompts)]
    want = tok(prompts, padding=True, truncation=True)
    assert np.array_equal(np.asarray(want["input_ids"]), got["input_ids"])
    assert np.array_equal(np.asarray(want["attention_mask"]), got["attention_mask"])


def _chunks_equal(a, b):
    assert a.n_requests == b.n_requests and list(a.counts) == list(b.counts)
    assert np.array_equal(np.asarray(a.ids), np.asarray(b.ids)) and np.asarray(a.lookup).tolist() == np.asarray(b.lookup).tolist()


def test_templated_path_deferred_probe(monkeypatch):
    """``defer_probe``: the chunk carries the longest row's cross-check against the public tokenizer call instead of running it;
    it says yes on an honest twin, and on a disagreement it says no and retires the twin (the engine then redoes the
    preparation on the generic path)."""
    from emcid_amd import compute_z as cz, host_text
    if not host_text.available():
        pytest.skip("libemcid_host.so not built")
    tok = syn.build_tokenizer(*syn.synthetic_vocab(syllables=True))
    if host_text.NativeClipBpe.for_tokenizer(tok) is None:
        pytest.skip("no native twin for the synthetic tokenizer")
    reqs = syn.make_requests(40, names="syllable", name_seed=77)
    now = cz.templated_prompt_chunk(tok, reqs, reqs[0])
    assert now is not None and now.verify is None
    reqs78 = syn.make_requests(40, names="syllable", name_seed=78)
    later = cz.templated_prompt_chunk(tok, reqs78, reqs[0], defer_probe=True)
    assert later is not None and callable(later.verify) and later.verify() is True and later.verify() is True
    _chunks_equal(later, cz.templated_prompt_chunk(tok, reqs78, reqs[0]))      # lookup positions from the construction == the walk's
    # a name that also occurs EARLIER in its prompt: the reference's walk stops at the first occurrence, the construction points at
    # the name itself -> the deferred check says no (the tokenizer twin stays: nothing is wrong with it), the up-front path is right
    nm = syn.syllable_names(3)
    twice = [{"source": nm[i], "dest": "x", "prompts": [f"art by {nm[0]} and {{}}", "style of {}"], "seed_train": 1} for i in range(3)]
    early = cz.templated_prompt_chunk(tok, twice, twice[0], defer_probe=True)
    sure = cz.templated_prompt_chunk(tok, twice, twice[0])
    assert early is not None and sure is not None and early.verify() is False
    assert host_text.NativeClipBpe.for_tokenizer(tok) is not None
    assert np.asarray(sure.lookup).tolist()[0] < np.asarray(early.lookup).tolist()[0]
    monkeypatch.setenv("EMCID_TEMPLATED", "0")
    _chunks_equal(sure, list(cz.iter_prompt_chunks(tok, twice, 1))[0])
    monkeypatch.setenv("EMCID_TEMPLATED", "1")
    third = cz.templated_prompt_chunk(tok, syn.make_requests(40, names="syllable", name_seed=79), reqs[0], defer_probe=True)
    real = type(tok).__call__

    def lying(self, prompts, **kw):
        out = real(self, prompts, **kw)
        out["input_ids"][0][1] += 1
        return out

    monkeypatch.setattr(type(tok), "__call__", lying)
    assert third.verify() is False
    monkeypatch.setattr(type(tok), "__call__", real)
    assert host_text.NativeClipBpe.for_tokenizer(tok) is None and cz.templated_prompt_chunk(tok, reqs, reqs[0]) is None
    chunks = list(cz.iter_prompt_chunks(tok, reqs, 1, defer_probe=True))          # generic path now: nothing left to verify
    assert len(chunks) == 1 and chunks[0].verify is None
    _chunks_equal(chunks[0], now)


def test_templated_prompt_path_equals_generic_path(monkeypatch):
    """templated_prompt_chunk (prefix/name/suffix encoded once, subjects passed once) against the generic path (every prompt
    formatted, tokenized and searched as a string): same ids, lookup positions, counts — mass-edit shape, ragged template
    sets, names with spaces / apostrophes / capitals, a name that also occurs inside a template, glued templates (no white
    space next to the braces), non-ASCII names (row-wise fallback to the HF tokenizer), 1-prompt requests."""
    from emcid_amd import compute_z as cz, host_text
    if not host_text.available():
        pytest.skip("libemcid_host.so not built")
    tok = syn.build_tokenizer(*syn.synthetic_vocab(syllables=True))
    if host_text.NativeClipBpe.for_tokenizer(tok) is None:
        pytest.skip("no native twin for the synthetic tokenizer")

    def both(reqs, n_chunks=1):
        monkeypatch.setenv("EMCID_TEMPLATED", "1")
        fast = list(cz.iter_prompt_chunks(tok, reqs, n_chunks))
        monkeypatch.setenv("EMCID_TEMPLATED", "0")
        slow = list(cz.iter_prompt_chunks(tok, reqs, n_chunks))
        assert len(fast) == len(slow)
        for a, b in zip(fast, slow):
            _chunks_equal(a, b)
        return fast

    reqs = syn.make_requests(300, names="syllable")
    assert cz.templated_prompt_chunk(tok, reqs, reqs[0]) is not None        # the fast path really serves the bench shape
    both(reqs)
    both(reqs, n_chunks=3)
    both(syn.make_requests(4500, names="syllable", name_seed=5))      # >= 4 096 names: the threaded name encoding of libemcid_host
    names = syn.syllable_names(12)
    odd = []
    for i, nm in enumerate(names):
        src = [nm, nm.upper(), f"{nm} {names[(i + 1) % 12]}", f"{nm}'s", f" {nm}", f"{nm} "][i % 6]
        tmpls = [["painting by {}", "{} style", "a{}b", "in the style of {} , oil"], ["{}"], ["art by {}", "{}, {}"[:2] + " art"],
                 [f"{nm} and {{}}"]][i % 4]
        odd.append({"source": src, "dest": "x", "prompts": tmpls, "seed_train": 1})
    both(odd)
    uni = [dict(r) for r in reqs[:9]]
    uni[4]["prompts"] = ["caf\u00e9 by {}", "{} caf\u00e9"]      # outside ASCII: those rows go through the HF tokenizer
    both(uni)
    uni[4]["source"] = "caf\u00e9 " + uni[4]["source"]          # a subject the synthetic vocabulary cannot spell: same error
    for flag in ("1", "0"):
        monkeypatch.setenv("EMCID_TEMPLATED", flag)
        with pytest.raises(ValueError, match="not found in tokens"):
            list(cz.iter_prompt_chunks(tok, uni, 1))
    # outside the templated shape: the generic path must serve these (and the fast one must decline)
    for bad in ([{"source": "ka", "dest": "x", "prompts": ["{0} art", "by {}"], "seed_train": 1}] * 4,
                [{"source": "ka", "dest": "x", "prompts": ["{{}} {}"], "seed_train": 1}] * 4,
                [{"source": "ka", "dest": "x", "source_prompts": ["art by ka"], "prompts": ["art by {}"], "seed_train": 1}] * 4):
        assert cz.templated_prompt_chunk(tok, bad, bad[0]) is None
        both(bad)
    with pytest.raises(ValueError):                              # subject not in the prompt: the scalar walk's error
        monkeypatch.setenv("EMCID_TEMPLATED", "1")
        list(cz.iter_prompt_chunks(tok, [{"source": "zu", "dest": "x", "prompts": ["art by ka {}"[:9]], "seed_train": 1}] * 4, 1))


def test_templated_prompt_path_randomised(monkeypatch):
    """Seeded random requests (names with spaces, capitals, apostrophes, digits, punctuation; templates with the braces glued to
    letters, digits, punctuation or white space; ragged template sets): wherever the templated path serves a request list it
    must give exactly what the generic path gives; where the generic path raises, so must it."""
    from emcid_amd import compute_z as cz, host_text
    if not host_text.available():
        pytest.skip("libemcid_host.so not built")
    tok = syn.build_tokenizer(*syn.synthetic_vocab(syllables=True))
    if host_text.NativeClipBpe.for_tokenizer(tok) is None:
        pytest.skip("no native twin for the synthetic tokenizer")
    rng = np.random.default_rng(11)
    syll = syn.syllable_names(40)
    glue = ["", " ", "  ", "-", "'s ", ", ", "7", ".", " the ", "\t"]

    def name():
        parts = [syll[int(rng.integers(len(syll)))] for _ in range(int(rng.integers(1, 4)))]
        s = glue[int(rng.integers(len(glue)))].join(parts) if rng.random() < 0.3 else " ".join(parts)
        if rng.random() < 0.2:
            s = s.title()
        if rng.random() < 0.1:
            s = s + "'s"
        return s.strip() or syll[0]

    def template():
        pre = glue[int(rng.integers(len(glue)))].join(syll[int(rng.integers(len(syll)))] for _ in range(int(rng.integers(0, 3))))
        suf = glue[int(rng.integers(len(glue)))].join(syll[int(rng.integers(len(syll)))] for _ in range(int(rng.integers(0, 3))))
        return pre + glue[int(rng.integers(len(glue)))] + "{}" + glue[int(rng.integers(len(glue)))] + suf

    served = 0
    for trial in range(25):
        sets = [[template() for _ in range(int(rng.integers(1, 4)))] for _ in range(int(rng.integers(1, 3)))]
        reqs = [{"source": name(), "dest": "x", "prompts": sets[int(rng.integers(len(sets)))], "seed_train": 1}
                for _ in range(int(rng.integers(3, 40)))]
        out = {}
        for flag in ("1", "0"):
            monkeypatch.setenv("EMCID_TEMPLATED", flag)
            try:
                out[flag] = list(cz.iter_prompt_chunks(tok, reqs, 1))
            except ValueError as e:
                out[flag] = ("ValueError", str(e)[:40])
        if isinstance(out["0"], tuple) or isinstance(out["1"], tuple):
            assert isinstance(out["0"], tuple) and isinstance(out["1"], tuple), (trial, out["0"] if isinstance(out["0"], tuple) else out["1"])
            continue
        served += cz.templated_prompt_chunk(tok, reqs, reqs[0]) is not None
        for a, b in zip(out["1"], out["0"]):
            _chunks_equal(a, b)
    assert served >= 15


# ---- round 3: native v* reader, lambda-free factor key, leased workspaces, opt-in thread management --------------------

def _vstar_setup(tmp_path, n=70, width=48):
    from emcid_amd import emcid_main as em
    reqs = syn.make_requests(n, names="syllable")
    cache = str(tmp_path / "cache") + "/"
    vs = syn.write_vstar_cache(cache, reqs, width, seed=1, scale=0.5)
    hp = EMCIDHyperParams(**syn.sd_hparams_dict(layers=(1, 2), mom2_update_weight=50, edit_weight=0.5))
    return em, reqs, cache, vs, hp


def test_native_vstar_reader_equals_numpy_reader(tmp_path, monkeypatch):
    """emcid_read_npz_rows_f32 (csrc/host_io.cpp) against np.load on the files np.savez writes (reference :951-968):
    identical rows; files it does not serve (compressed member, other first member) are flagged and read by numpy, a
    float64 / (1, width) file is converted like astype; a missing file goes to stage1; any thread count gives the same rows."""
    em, reqs, cache, vs, hp = _vstar_setup(tmp_path)
    name = lambda i: em.vstar_cache_name(cache, reqs[i], hp, i)
    np.savez_compressed(name(3), v_star=vs[3])
    np.savez(name(4), v_star=vs[4].astype(np.float64))
    np.savez(name(5), v_star=vs[5][None])
    np.savez(name(6), other=np.zeros(3), v_star=vs[6])
    Path(name(7)).unlink()
    monkeypatch.setenv("EMCID_NATIVE_VSTAR", "0")
    ref = em.load_v_stars(reqs, hp, cache, stage1=lambda r, sfx: torch.full((48,), 7.0))
    Path(name(7)).unlink()          # the numpy pass wrote Stage 1's result to the cache; make it a miss again
    monkeypatch.setenv("EMCID_NATIVE_VSTAR", "1")
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("EMCID_READ_THREADS", threads)
        rows, status = em._native_vstar_rows([name(i) for i in range(len(reqs))], 48, False)
        assert list(status[:9]) == [0, 0, 0, 2, 0, 0, 2, 1, 0] and not status[9:].any()
        keep = status == 0
        assert np.array_equal(rows.numpy()[keep], ref.numpy()[keep])
    got = em.load_v_stars(reqs, hp, cache, stage1=lambda r, sfx: torch.full((48,), 7.0), width=48)
    assert torch.equal(got, ref) and torch.equal(got[7], torch.full((48,), 7.0))
    assert Path(name(7)).exists()                                   # Stage 1's v* was written to the cache (:951-968)
    # a wrong width is "not such a file": numpy reads it and the engine's own shape check reports it
    rows, status = em._native_vstar_rows([name(0)], 32, False)
    assert list(status) == [2]
    # without a width (callers that do not know the hidden size) the per-file path serves everything
    assert torch.equal(em.load_v_stars(reqs, hp, cache), ref)


def test_early_vstar_reader_survives_dropped_plans_and_falls_back(tmp_path):
    """emcid_main._EarlyVstars: the native read runs on a helper thread from the moment it is started; a handle that is dropped
    unconsumed (a plan prepared and never run) must not free the buffers under the reader (they belong to the reader's task), the
    rows equal load_v_stars's, and whatever the native reader does not serve (a miss, a file for numpy) takes load_v_stars at
    result()."""
    import gc
    em, reqs, cache, vs, hp = _vstar_setup(tmp_path, n=400, width=64)
    ref = em.load_v_stars(reqs, hp, cache, width=64)
    for _ in range(30):                                   # started and dropped at once, many times over
        em._EarlyVstars.start(reqs, hp, cache, "", None, width=64, pin=False)
    gc.collect()
    early = em._EarlyVstars.start(reqs, hp, cache, "", None, width=64, pin=False)
    assert early is not None and torch.equal(early.result(), ref)
    name = lambda i: em.vstar_cache_name(cache, reqs[i], hp, i)
    np.savez_compressed(name(3), v_star=vs[3])            # not served natively: the whole list goes through load_v_stars
    Path(name(7)).unlink()
    early = em._EarlyVstars.start(reqs, hp, cache, "", lambda r, sfx: torch.full((64,), 7.0), width=64, pin=False)
    # ... with the rows the batch read DID serve handed on: no second native read of the 398 hits (round-4 advisor, low)
    calls = []
    real = em._native_vstar_rows
    em._native_vstar_rows = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        got = early.result()
    finally:
        em._native_vstar_rows = real
    assert not calls
    assert torch.equal(got[3], ref[3]) and torch.equal(got[7], torch.full((64,), 7.0)) and torch.equal(got[8:], ref[8:])
    assert em._EarlyVstars.start(reqs, hp, None, "", None, width=64) is None          # nothing to read early without a cache


def test_vstar_memo_is_bounded(tmp_path, monkeypatch):
    em, reqs, cache, vs, hp = _vstar_setup(tmp_path, n=12)
    monkeypatch.setattr(em, "_VSTAR_CACHE_MAX", 5)
    em._VSTAR_CACHE.clear()
    monkeypatch.setenv("EMCID_NATIVE_VSTAR", "0")
    em.load_v_stars(reqs, hp, cache)
    assert len(em._VSTAR_CACHE) == 5


def test_factor_cache_key_ignores_lambda_but_not_edit_weight():
    from emcid_amd.edit_engine import factor_cache_key
    c = [torch.zeros(4, 4), torch.ones(4, 4)]
    assert factor_cache_key(c, 4000.0, 0.5) == factor_cache_key(c, 123.0, 0.5)
    assert factor_cache_key(c, 4000.0, 0.5) != factor_cache_key(c, 4000.0, 0.6)
    assert factor_cache_key(c, 4000.0, 0.5) != factor_cache_key(c[::-1], 4000.0, 0.5)
    c[0].add_(1.0)                                  # statistics changed in place: version counter moves
    k2 = factor_cache_key(c, 4000.0, 0.5)
    c[0].add_(1.0)
    assert k2 != factor_cache_key(c, 4000.0, 0.5)


def test_lam_ratio_rules():
    f = hip.CovFactors.__new__(hip.CovFactors)
    f.lam = 4000.0
    assert f.lam_ratio(None) == 1.0 and f.lam_ratio(4000) == 1.0 and f.lam_ratio(1000.0) == 0.25
    with pytest.raises(hip.EmcidHipError):
        f.lam_ratio(0.0)
    f.lam = None
    assert f.lam_ratio(5.0) == 1.0


def test_workspaces_are_leased_per_plan(monkeypatch):
    """Two plans of one shape in flight never share a workspace (buffers + info word); a plan gets its own back; a lease ends
    at check_info or with the plan."""
    import gc
    from types import SimpleNamespace
    from emcid_amd import edit_engine as ee

    class FakeWs:
        made = 0

        def __init__(self, N, d, h, dev):
            FakeWs.made += 1
            self.key = (N, d, h)

    monkeypatch.setattr(hip, "DualWorkspace", FakeWs)
    ee.clear_engine_caches()

    class Plan(SimpleNamespace):
        pass

    a, b = Plan(ws=None, dual_ws=None), Plan(ws=None, dual_ws=None)
    a.dual_ws = ee._workspace("dual", 10, 128, 32, "cuda:0", a)
    b.dual_ws = ee._workspace("dual", 10, 128, 32, "cuda:0", b)
    assert a.dual_ws is not b.dual_ws and FakeWs.made == 2
    assert ee._workspace("dual", 10, 128, 32, "cuda:0", a) is a.dual_ws          # a second run of the same plan
    ee._release_workspaces(a)                                                    # what check_info does
    c = Plan(ws=None, dual_ws=None)
    c.dual_ws = ee._workspace("dual", 10, 128, 32, "cuda:0", c)
    assert c.dual_ws is a.dual_ws and FakeWs.made == 2
    held = b.dual_ws
    del b
    gc.collect()                                                                 # a plan that was never checked
    d_ = Plan(ws=None, dual_ws=None)
    assert ee._workspace("dual", 10, 128, 32, "cuda:0", d_) is held
    e = Plan(ws=None, dual_ws=None)
    extra = ee._workspace("dual", 10, 128, 32, "cuda:0", e)                      # both pooled ones leased: a third, unpooled
    assert extra is not held and extra is not c.dual_ws and FakeWs.made == 3
    ee.clear_engine_caches()


def test_import_has_no_thread_side_effects():
    """`import emcid_amd` leaves the host application's thread pools and environment alone; manage_threads() is opt-in."""
    import subprocess
    import sys
    code = ("import os, torch\n"
            "os.environ.pop('RAYON_NUM_THREADS', None); os.environ.pop('EMCID_MANAGE_THREADS', None)\n"
            "torch.set_num_threads(3); before = torch.get_num_threads()\n"
            "import emcid_amd, emcid_amd.edit_engine, emcid_amd.emcid_main\n"
            "assert torch.get_num_threads() == before and 'RAYON_NUM_THREADS' not in os.environ\n"
            "assert emcid_amd.manage_threads() is False\n"
            "assert emcid_amd.manage_threads(force=True) is True and emcid_amd.manage_threads(force=True) is False\n"
            "assert torch.get_num_threads() <= before\n"
            "print('OK')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(REPO), timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


def test_host_library_exports_every_declared_symbol():
    from emcid_amd import host_text
    header = (REPO / "include" / "emcid_host.h").read_text()
    declared = set(re.findall(r"\b(emcid_[a-z0-9_]+)\s*\(", header))
    lib = ctypes.CDLL(str(host_text.lib_path()))
    for name in declared:
        assert hasattr(lib, name), name
    assert host_text.load().emcid_host_abi_version() == host_text.ABI_VERSION
    assert lib.emcid_read_npz_rows_f32(None, None, 1, b"v_star", 4, None, 4, None, 1) == -1


def test_invalidate_weight_caches_forgets_the_graph():
    """clip_forward.invalidate_weight_caches drops the per-encoder graph (snapshots, planes, native structs) so that the next
    call re-derives them — the hook for code that rewrites weights behind the version counter's back."""
    from emcid_amd import clip_forward as cf, synthetic as syn
    pipe = syn.build_pipe("toy", "cpu")
    g1 = cf.discover_cached(pipe.text_encoder, "text_model.encoder.layers.{}")
    assert cf.discover_cached(pipe.text_encoder, "text_model.encoder.layers.{}") is g1
    cf.invalidate_weight_caches(pipe.text_encoder)
    g2 = cf.discover_cached(pipe.text_encoder, "text_model.encoder.layers.{}")
    assert g2 is not g1
    cf.invalidate_weight_caches()
    assert cf.discover_cached(pipe.text_encoder, "text_model.encoder.layers.{}") is not g2
