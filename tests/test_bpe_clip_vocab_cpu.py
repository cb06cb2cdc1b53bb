"""The native CLIP BPE twin (csrc/host_text.cpp) against the HF tokenizer on a CLIP-SHAPED vocabulary.

The reference runs SD-v1.4's 49 408-entry byte-level CLIP BPE (scripts/run_emcid.py:63-76 loads the pipeline's tokenizer;
experiments/causal_trace.py:1057-1103 walks its tokens).  Neither `vocab.json` nor `merges.txt` exists offline, and the
benchmark's synthetic vocabulary has ~200 ASCII entries: no byte-level alphabet beyond ASCII, no deep merge chains.  Here a
vocabulary of the real SIZE and KIND is trained on the spot — the 256-symbol byte-level alphabet, each symbol with and without
`</w>`, and 48 894 merges learned by the `tokenizers` BPE trainer from a synthetic corpus (Zipf-distributed pseudo-words of 1-5
syllables, accented and non-Latin words, digits, punctuation runs) through CLIP's own normalizer and pre-tokenizer — and the
twin is fuzzed against `CLIPTokenizer` built from it: mixed ASCII / accented / apostrophe / digit / symbol prompts, the subject
token-range walk, and the encode throughput of 3 000 prompts (printed)."""
import json
import time

import numpy as np
import pytest

from emcid_amd import causal_trace, host_text, synthetic as syn

CONS, VOW = "bcdfghjklmnprstvwz", "aeiou"
ACCENTED = ["é", "è", "ñ", "ü", "ö", "ç", "å", "ø", "ł", "ń", "š", "ž", "ı", "ğ", "á", "í", "ó", "ú", "â", "ê"]
OTHER = ["привет", "мир", "художник", "日本", "絵", "画家", "ελληνικά", "τέχνη", "שלום", "فن"]


def _lexicon(rng, n):
    words, seen = [], set()
    while len(words) < n:
        k = int(rng.choice([1, 2, 3, 4, 5], p=[0.08, 0.32, 0.34, 0.18, 0.08]))
        w = "".join(CONS[rng.integers(len(CONS))] + VOW[rng.integers(len(VOW))] + (CONS[rng.integers(len(CONS))] if rng.random() < 0.35 else "")
                    for _ in range(k))
        if rng.random() < 0.08:
            i = int(rng.integers(len(w)))
            w = w[:i] + ACCENTED[rng.integers(len(ACCENTED))] + w[i + 1:]
        if w not in seen:
            seen.add(w)
            words.append(w)
    return words


@pytest.fixture(scope="module")
def clip_shaped():
    """(tokenizer, lexicon): a CLIPTokenizer whose vocabulary has CLIP's size and structure, trained in a few seconds."""
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, trainers, Regex

    rng = np.random.default_rng(7)
    lex = _lexicon(rng, 120000) + OTHER
    ranks = np.arange(1, len(lex) + 1, dtype=np.float64)
    p = 1.0 / ranks ** 0.9
    p /= p.sum()
    draws = rng.choice(len(lex), size=1_500_000, p=p)
    lines = []
    for i in range(0, len(draws), 12):
        ws = [lex[j] for j in draws[i:i + 12]]
        if i % 5 == 0:
            ws.insert(3, str(int(rng.integers(0, 100000))))
        if i % 7 == 0:
            ws[1] = ws[1] + "'s"
        if i % 11 == 0:
            ws.insert(5, rng.choice(["--", "...", "!!", "(", ")", ",", ";", "&", "#", "@", "%"]))
        lines.append(" ".join(ws))
    base = json.loads(syn.build_tokenizer()._tokenizer.to_str())
    split = base["pre_tokenizer"]["pretokenizers"][0]["pattern"]["Regex"]
    t = Tokenizer(models.BPE(end_of_word_suffix="</w>", unk_token=None))
    t.normalizer = normalizers.Sequence([normalizers.NFC(), normalizers.Replace(Regex(r"\s+"), " "), normalizers.Lowercase()])
    t.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split(Regex(split), behavior="removed", invert=True),
                                               pre_tokenizers.ByteLevel(add_prefix_space=False)])
    alphabet = pre_tokenizers.ByteLevel.alphabet()
    trainer = trainers.BpeTrainer(vocab_size=49406, initial_alphabet=alphabet, end_of_word_suffix="</w>", special_tokens=[],
                                  show_progress=False, min_frequency=1)
    t.train_from_iterator(lines, trainer)
    model = json.loads(t.to_str())["model"]
    vocab = dict(model["vocab"])
    merges = [tuple(m.split(" ")) if isinstance(m, str) else tuple(m) for m in model["merges"]]
    for a in alphabet:                     # CLIP's vocabulary lists every byte symbol with and without the end-of-word suffix
        for s in (a, a + "</w>"):
            if s not in vocab:
                vocab[s] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    tok = syn.build_tokenizer(vocab, merges)
    return tok, lex


def _prompts(rng, lex, n):
    out = []
    templates = ["a painting by {}", "an image of {} in the style of {}", "{}", "{}'s {} , {} !", "the {} -- {} ( {} )", "{} {} {} {} {}"]
    specials = ["don't", "it's", "we're", "rock'n'roll", "’", "“quoted”", "naïve café", "Łódź", "Ünïcödé", "x1y22z 007", "100% #1 @home $5",
                "a.b.c...", "tabs\tand\nnewlines", "  spaced   out  ", "MiXeD CaSe", "emoji 🎨 art", "日本 の 絵", "привет мир", ""]
    for i in range(n):
        tpl = templates[int(rng.integers(len(templates)))]
        ws = [lex[int(rng.integers(len(lex)))] if rng.random() < 0.9 else specials[int(rng.integers(len(specials)))]
              for _ in range(tpl.count("{}"))]
        if rng.random() < 0.1:
            ws = [w.upper() for w in ws]
        s = tpl.format(*ws)
        if rng.random() < 0.03:
            s = " ".join([s] * 12)            # longer than the 77-token window
        out.append(s)
    return out


def test_clip_shaped_vocabulary_has_the_real_size_and_depth(clip_shaped):
    tok, lex = clip_shaped
    cfg = json.loads(tok._tokenizer.to_str())
    assert len(cfg["model"]["vocab"]) == 49408 or abs(len(cfg["model"]["vocab"]) - 49408) <= 512
    assert len(cfg["model"]["merges"]) >= 45000
    # deep chains: frequent words are single tokens made by >= 4 merges, rare ones split into several
    lens = [len(tok.tokenize(w)) for w in lex[:200]]
    assert min(lens) == 1 and np.mean([len(tok.tokenize(w)) for w in lex[-200:]]) > 1.5


def test_native_twin_agrees_with_hf_on_a_clip_shaped_vocabulary(clip_shaped):
    tok, lex = clip_shaped
    if not host_text.available():
        pytest.skip("libemcid_host.so not built")
    twin = host_text.NativeClipBpe.for_tokenizer(tok)
    assert twin is not None, "the native twin refused a CLIP pipeline / failed its probes"
    rng = np.random.default_rng(11)
    prompts = _prompts(rng, lex, 3000)
    want = tok(prompts, padding=True, truncation=True)
    t0 = time.perf_counter()
    ids, lengths, fb = twin.encode(prompts)
    dt = time.perf_counter() - t0
    got = twin.tokenize(tok, prompts)
    assert np.array_equal(np.asarray(want["input_ids"], dtype=np.int64), got["input_ids"])
    assert np.array_equal(np.asarray(want["attention_mask"], dtype=np.int64), got["attention_mask"])
    # rows the twin does itself (no fallback flag) are already the HF rows; pure-ASCII prompts never fall back
    S = got["input_ids"].shape[1]
    own = ~fb
    assert np.array_equal(ids[own][:, :S], got["input_ids"][own])
    ascii_rows = np.array([p.isascii() for p in prompts])
    assert not fb[ascii_rows].any()
    print(f"\nnative BPE on a CLIP-shaped vocabulary ({len(json.loads(tok._tokenizer.to_str())['model']['merges'])} merges): "
          f"{len(prompts)} prompts in {dt * 1e3:.1f} ms = {len(prompts) / dt / 1e3:.0f} k prompts/s, {int(fb.sum())} rows through the HF "
          f"fallback ({int((~ascii_rows).sum())} non-ASCII prompts)")
    t0 = time.perf_counter()
    tok(prompts, padding=True, truncation=True)
    print(f"HF tokenizer on the same prompts: {(time.perf_counter() - t0) * 1e3:.1f} ms")


def test_subject_token_ranges_on_a_clip_shaped_vocabulary(clip_shaped):
    """find_token_range (reference experiments/causal_trace.py:1057-1103): the batch walk — which goes through libemcid_host for
    the rows it can serve — against the reference-faithful scalar walk on multi-token subjects of a CLIP-shaped vocabulary, with
    accents, apostrophes and non-Latin text among them."""
    tok, lex = clip_shaped
    rng = np.random.default_rng(13)
    subjects = [lex[int(rng.integers(1000, len(lex)))] + " " + lex[int(rng.integers(1000, len(lex)))] for _ in range(300)]
    subjects += ["d'artagnan", "o'keeffe", "van gogh", "naïve café", "Łódź mevni"]
    prompts = [f"a painting by {s}" if i % 3 else f"{s} , in the style of art" for i, s in enumerate(subjects)]
    enc = tok(prompts, padding=True, truncation=True)
    ids = np.asarray(enc["input_ids"], dtype=np.int64)
    walker = causal_trace.TokenRangeFinder(tok)
    scalar = []
    for i, s in enumerate(subjects):
        scalar.append(walker(ids[i].tolist(), s))
    assert walker.batch(ids, subjects) == scalar
    last = walker.last_tokens(ids, subjects, np.arange(len(subjects)))
    assert last.tolist() == [b - 1 for _, b in scalar]
    n_tok = np.asarray(enc["attention_mask"]).sum(1)
    for i, (a, b) in enumerate(scalar):
        assert 0 < a < b < n_tok[i]                                  # inside the prompt, behind the start token, before the end token
        if subjects[i].isascii():
            spelled = tok.decode(ids[i, a:b].tolist()).replace(" ", "")
            assert subjects[i].lower().replace(" ", "") in spelled    # the range's tokens spell (at least) the subject
        # (a character whose bytes are split over two tokens counts once in the decoded prompt and twice in the per-token walk:
        #  the reference's own arithmetic, special-cased there for one letter only — kept as it is, batch == scalar above)
    multi = sum(1 for a, b in scalar if b - a >= 3)
    assert multi > 100                                                # real multi-token subjects, not one token per word
