// host_text.cpp — libemcid_host.so: CLIP byte-level BPE encoding and the subject token-range walk on the host
// (include/emcid_host.h).  Plain C++17, no GPU.  Restates, for ASCII prompts, what the reference gets from the Hugging Face
// CLIP tokenizer (emcid/compute_z.py:65) and from find_token_range (experiments/causal_trace.py:1057); everything else is
// flagged back to the caller, prompt by prompt.
#include "../../include/emcid_host.h"

#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

thread_local std::string g_error;

inline bool is_space(unsigned char c) { return c == ' ' || (c >= 0x09 && c <= 0x0d); }
inline bool is_letter(unsigned char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }
inline bool is_digit(unsigned char c) { return c >= '0' && c <= '9'; }
inline bool servable(unsigned char c) { return is_space(c) || (c >= 0x21 && c <= 0x7e); }

struct Merge {
    int32_t rank, id;
};

// length of the contraction piece starting at s[p] (lower-cased text), 0 if there is none
inline size_t contraction(const std::string& s, size_t p, size_t len) {
    if (s[p] != '\'' || p + 1 >= len) return 0;
    const char c1 = s[p + 1], c2 = p + 2 < len ? s[p + 2] : '\0';
    if (c1 == 's' || c1 == 't') return 2;
    if ((c1 == 'r' && c2 == 'e') || (c1 == 'v' && c2 == 'e')) return 3;
    if (c1 == 'm') return 2;
    if (c1 == 'l' && c2 == 'l') return 3;
    if (c1 == 'd') return 2;
    return 0;
}

}  // namespace

struct emcid_bpe {
    int32_t sym[128];       // id of the one-character token, -1 = not in the vocabulary
    int32_t sym_end[128];   // id of character + end-of-word suffix
    std::unordered_map<uint64_t, Merge> merges;
    std::unordered_map<std::string, std::vector<int32_t>> cache;    // pre-token -> ids
    std::mutex lock;

    static uint64_t key(int32_t a, int32_t b) { return ((uint64_t)(uint32_t)a << 32) | (uint32_t)b; }

    // The ids of one pre-token (lower-case printable ASCII, no spaces); false if a character is not in the vocabulary.
    // tokenizers' BPE: start from characters (the last one carries the suffix), repeatedly apply the applicable merge of
    // lowest rank, leftmost first.
    bool word(const char* w, size_t n, const std::vector<int32_t>** out) {
        std::string k(w, n);
        auto it = cache.find(k);
        if (it != cache.end()) {
            *out = &it->second;
            return true;
        }
        std::vector<int32_t> s(n);
        for (size_t i = 0; i < n; ++i) {
            const unsigned char c = (unsigned char)w[i];
            s[i] = (i + 1 == n) ? sym_end[c] : sym[c];
            if (s[i] < 0) return false;
        }
        while (s.size() > 1) {
            int32_t best = INT32_MAX, at = -1, id = -1;
            for (size_t i = 0; i + 1 < s.size(); ++i) {
                auto m = merges.find(key(s[i], s[i + 1]));
                if (m != merges.end() && m->second.rank < best) best = m->second.rank, at = (int32_t)i, id = m->second.id;
            }
            if (at < 0) break;
            s[at] = id;
            s.erase(s.begin() + at + 1);
        }
        if (cache.size() > (1u << 20)) cache.clear();
        *out = &(cache[k] = std::move(s));
        return true;
    }
};

extern "C" {

int emcid_host_abi_version(void) { return 1; }

const char* emcid_host_last_error(void) { return g_error.c_str(); }

emcid_bpe* emcid_bpe_create(const char* vocab_bytes, const int64_t* vocab_off, const int32_t* vocab_ids, int64_t n_vocab,
                            const int32_t* merges, int64_t n_merges, const char* end_of_word_suffix) {
    if (!vocab_bytes || !vocab_off || !vocab_ids || n_vocab <= 0 || n_merges < 0 || (n_merges && !merges) || !end_of_word_suffix) {
        g_error = "emcid_bpe_create: bad argument";
        return nullptr;
    }
    std::unordered_map<std::string, int32_t> vocab;
    std::unordered_map<int32_t, std::string> text;
    vocab.reserve((size_t)n_vocab * 2);
    text.reserve((size_t)n_vocab * 2);
    for (int64_t i = 0; i < n_vocab; ++i) {
        if (vocab_off[i + 1] < vocab_off[i]) {
            g_error = "emcid_bpe_create: vocabulary offsets are not monotone";
            return nullptr;
        }
        std::string t(vocab_bytes + vocab_off[i], (size_t)(vocab_off[i + 1] - vocab_off[i]));
        vocab[t] = vocab_ids[i];
        text[vocab_ids[i]] = std::move(t);
    }
    auto* m = new emcid_bpe();
    const std::string suffix(end_of_word_suffix);
    for (int c = 0; c < 128; ++c) {
        m->sym[c] = m->sym_end[c] = -1;
        if (c < 0x21 || c > 0x7e) continue;     // ByteLevel maps printable ASCII to itself
        const std::string ch(1, (char)c);
        auto a = vocab.find(ch);
        if (a != vocab.end()) m->sym[c] = a->second;
        auto b = vocab.find(ch + suffix);
        if (b != vocab.end()) m->sym_end[c] = b->second;
    }
    m->merges.reserve((size_t)n_merges * 2);
    for (int64_t r = 0; r < n_merges; ++r) {
        const int32_t a = merges[2 * r], b = merges[2 * r + 1];
        auto ta = text.find(a), tb = text.find(b);
        if (ta == text.end() || tb == text.end()) {
            g_error = "emcid_bpe_create: merge " + std::to_string(r) + " names an id outside the vocabulary";
            delete m;
            return nullptr;
        }
        auto merged = vocab.find(ta->second + tb->second);
        if (merged == vocab.end()) {
            g_error = "emcid_bpe_create: merge " + std::to_string(r) + " produces a token outside the vocabulary";
            delete m;
            return nullptr;
        }
        m->merges.emplace(emcid_bpe::key(a, b), Merge{(int32_t)r, merged->second});    // first (lowest) rank wins
    }
    return m;
}

void emcid_bpe_destroy(emcid_bpe* m) { delete m; }

int64_t emcid_bpe_encode_batch(emcid_bpe* m, const char* text, const int64_t* off, int64_t n, int32_t bos, int32_t eos,
                               int32_t pad, int32_t max_len, int64_t* ids, int32_t* lengths, uint8_t* fallback) {
    if (!m || !text || !off || n < 0 || max_len < 2 || !ids || !lengths || !fallback) {
        g_error = "emcid_bpe_encode_batch: bad argument";
        return -1;
    }
    std::lock_guard<std::mutex> guard(m->lock);
    int64_t n_fallback = 0;
    std::string low;
    std::vector<int32_t> row;
    for (int64_t i = 0; i < n; ++i) {
        int64_t* out = ids + i * (int64_t)max_len;
        for (int32_t j = 0; j < max_len; ++j) out[j] = pad;
        lengths[i] = 0;
        fallback[i] = 0;
        if (off[i + 1] < off[i]) {
            g_error = "emcid_bpe_encode_batch: text offsets are not monotone";
            return -1;
        }
        const char* s = text + off[i];
        const size_t len = (size_t)(off[i + 1] - off[i]);
        bool ok = true;
        low.assign(s, len);
        for (size_t p = 0; p < len && ok; ++p) {
            const unsigned char c = (unsigned char)low[p];
            if (c >= 0x80 || !servable(c) || (c == '<' && p + 1 < len && low[p + 1] == '|')) ok = false;
            if (c >= 'A' && c <= 'Z') low[p] = (char)(c + 32);
        }
        row.clear();
        size_t p = 0;
        const int32_t budget = max_len - 2;
        while (ok && p < len) {
            const unsigned char c = (unsigned char)low[p];
            if (is_space(c)) {
                ++p;
                continue;
            }
            size_t q = p + contraction(low, p, len);    // 's|'t|'re|'ve|'m|'ll|'d come first in the alternation
            if (q == p) {
                q = p + 1;
                if (is_letter(c)) {                         // \p{L}+
                    while (q < len && is_letter((unsigned char)low[q])) ++q;
                } else if (!is_digit(c)) {                  // [^\s\p{L}\p{N}]+   (a digit is a piece of its own: \p{N})
                    while (q < len && !is_space((unsigned char)low[q]) && !is_letter((unsigned char)low[q]) &&
                           !is_digit((unsigned char)low[q]))
                        ++q;
                }
            }
            const std::vector<int32_t>* w = nullptr;
            if (!m->word(low.data() + p, q - p, &w)) {
                ok = false;
                break;
            }
            row.insert(row.end(), w->begin(), w->end());
            p = q;
            if ((int32_t)row.size() >= budget) break;    // truncation keeps a prefix: later pieces cannot matter
        }
        if (!ok) {
            fallback[i] = 1;
            ++n_fallback;
            continue;
        }
        const int32_t keep = (int32_t)row.size() < budget ? (int32_t)row.size() : budget;
        out[0] = bos;
        for (int32_t j = 0; j < keep; ++j) out[1 + j] = row[j];
        out[1 + keep] = eos;
        lengths[i] = keep + 2;
    }
    return n_fallback;
}

int64_t emcid_find_token_ranges(const int64_t* ids, int64_t n, int64_t S, const char* piece_ns, const int64_t* piece_off,
                                const int32_t* piece_len, int64_t n_pieces, const char* subj, const int64_t* subj_off,
                                const char* forbid, int32_t* first, int32_t* last, uint8_t* status) {
    if (!ids || n < 0 || S <= 0 || !piece_ns || !piece_off || !piece_len || n_pieces <= 0 || !subj || !subj_off || !first ||
        !last || !status) {
        g_error = "emcid_find_token_ranges: bad argument";
        return -1;
    }
    int64_t n_scalar = 0;
    std::string whole;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t* row = ids + i * S;
        first[i] = last[i] = 0;
        status[i] = 1;
        const size_t sl = (size_t)(subj_off[i + 1] - subj_off[i]);
        bool ok = sl > 0 && subj_off[i + 1] >= subj_off[i];
        whole.clear();
        for (int64_t j = 0; j < S && ok; ++j) {
            const int64_t t = row[j];
            if (t < 0 || t >= n_pieces || piece_len[t] < 0) ok = false;
            else whole.append(piece_ns + piece_off[t], (size_t)(piece_off[t + 1] - piece_off[t]));
        }
        if (ok && forbid && *forbid && whole.find(forbid) != std::string::npos) ok = false;
        if (ok) {
            const size_t at = whole.find(subj + subj_off[i], 0, sl);
            if (at == std::string::npos) ok = false;
            else {
                // the reference's walk: `seen` counts len(decode([t])) (spaces included) against offsets in the space-free
                // string; first = first token with seen > at, last = one past the first token with seen >= at + len(sub)
                const int64_t stop = (int64_t)(at + sl);
                int64_t seen = 0;
                int32_t f = -1, l = -1;
                for (int64_t j = 0; j < S; ++j) {
                    seen += piece_len[row[j]];
                    if (f < 0 && seen > (int64_t)at) f = (int32_t)j;
                    if (seen >= stop) {
                        l = (int32_t)j + 1;
                        break;
                    }
                }
                if (f >= 0 && l >= 0) first[i] = f, last[i] = l, status[i] = 0;
                else ok = false;
            }
        }
        if (!ok) ++n_scalar;
    }
    return n_scalar;
}

}  // extern "C"
