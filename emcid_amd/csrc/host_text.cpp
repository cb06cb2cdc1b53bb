// host_text.cpp — libemcid_host.so: CLIP byte-level BPE encoding and the subject token-range walk on the host
// (include/emcid_host.h).  Plain C++17, no GPU.  Restates, for ASCII prompts, what the reference gets from the Hugging Face
// CLIP tokenizer (emcid/compute_z.py:65) and from find_token_range (experiments/causal_trace.py:1057); everything else is
// flagged back to the caller, prompt by prompt.
#include "../../include/emcid_host.h"

#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <atomic>
#include <cstdlib>
#include <thread>
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

// (left id, right id) -> merge: open addressing on a power-of-two table at most half full, multiplicative hash — merge_word asks
// ~25 times per never-seen name, and a node-based std::unordered_map answered in ~40 ns where this takes ~8
struct FlatMerges {
    struct Slot { uint64_t key; Merge m; };
    std::vector<Slot> slots;
    int shift = 64;
    static constexpr uint64_t EMPTY = ~0ull;
    void reserve(size_t n) {
        size_t cap = 16;
        int bits = 4;
        while (cap < 2 * n + 2) cap <<= 1, ++bits;
        slots.assign(cap, Slot{EMPTY, Merge{0, 0}});
        shift = 64 - bits;
    }
    inline size_t home(uint64_t key) const { return (size_t)((key * 0x9E3779B97F4A7C15ull) >> shift); }
    void insert_if_absent(uint64_t key, Merge m) {
        const size_t mask = slots.size() - 1;
        for (size_t i = home(key);; i = (i + 1) & mask) {
            if (slots[i].key == key) return;
            if (slots[i].key == EMPTY) {
                slots[i] = Slot{key, m};
                return;
            }
        }
    }
    inline const Merge* find(uint64_t key) const {
        const size_t mask = slots.size() - 1;
        for (size_t i = home(key);; i = (i + 1) & mask) {
            if (slots[i].key == key) return &slots[i].m;
            if (slots[i].key == EMPTY) return nullptr;
        }
    }
};

struct emcid_bpe {
    int32_t sym[128];       // id of the one-character token, -1 = not in the vocabulary
    int32_t sym_end[128];   // id of character + end-of-word suffix
    FlatMerges merges;
    std::unordered_map<std::string, std::vector<int32_t>> cache;    // pre-token -> ids
    std::mutex lock;

    static uint64_t key(int32_t a, int32_t b) { return ((uint64_t)(uint32_t)a << 32) | (uint32_t)b; }

    // The ids of one pre-token (lower-case printable ASCII, no spaces); false if a character is not in the vocabulary.
    // tokenizers' BPE: start from characters (the last one carries the suffix), repeatedly apply the applicable merge of
    // lowest rank, leftmost first.  Pure (reads the model only): safe from several threads at once.
    bool merge_word(const char* w, size_t n, std::vector<int32_t>& s) const {
        s.resize(n);
        for (size_t i = 0; i < n; ++i) {
            const unsigned char c = (unsigned char)w[i];
            s[i] = (i + 1 == n) ? sym_end[c] : sym[c];
            if (s[i] < 0) return false;
        }
        size_t len = n;
        int32_t* v = s.data();
        while (len > 1) {
            int32_t best = INT32_MAX, at = -1, id = -1;
            for (size_t i = 0; i + 1 < len; ++i) {
                const Merge* m = merges.find(key(v[i], v[i + 1]));
                if (m != nullptr && m->rank < best) best = m->rank, at = (int32_t)i, id = m->id;
            }
            if (at < 0) break;
            v[at] = id;
            for (size_t i = (size_t)at + 1; i + 1 < len; ++i) v[i] = v[i + 1];
            --len;
        }
        s.resize(len);
        return true;
    }

    // merge_word through the pre-token cache (caller holds `lock`)
    bool word(const char* w, size_t n, const std::vector<int32_t>** out) {
        std::string k(w, n);
        auto it = cache.find(k);
        if (it != cache.end()) {
            *out = &it->second;
            return true;
        }
        std::vector<int32_t> s;
        if (!merge_word(w, n, s)) return false;
        if (cache.size() > (1u << 20)) cache.clear();
        *out = &(cache[k] = std::move(s));
        return true;
    }
};

extern "C" {

int emcid_host_abi_version(void) { return 5; }

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
    m->merges.reserve((size_t)n_merges);
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
        m->merges.insert_if_absent(emcid_bpe::key(a, b), Merge{(int32_t)r, merged->second});    // first (lowest) rank wins
    }
    return m;
}

void emcid_bpe_destroy(emcid_bpe* m) { delete m; }

}  // extern "C"  (C++ helpers with overloads / templates)

// ids of one piece without a heap block for the usual few tokens (a mass edit encodes ~1 000 never-seen names per call; a
// std::vector per name and per pre-token was most of that call's time: four allocations per name against ~15 table lookups)
struct SmallIds {
    int32_t local[16];
    int32_t n = 0;
    std::vector<int32_t> heap;          // used once the piece outgrows `local`
    size_t size() const { return heap.empty() ? (size_t)n : heap.size(); }
    bool empty() const { return size() == 0; }
    const int32_t* begin() const { return heap.empty() ? local : heap.data(); }
    const int32_t* end() const { return begin() + size(); }
    void append(const int32_t* b, const int32_t* e) {
        const size_t k = (size_t)(e - b);
        if (heap.empty() && (size_t)n + k <= 16) {
            for (size_t i = 0; i < k; ++i) local[n + (int32_t)i] = b[i];
            n += (int32_t)k;
            return;
        }
        if (heap.empty()) heap.assign(local, local + n);
        heap.insert(heap.end(), b, e);
    }
};
static inline void ids_append(std::vector<int32_t>& row, const int32_t* b, const int32_t* e) { row.insert(row.end(), b, e); }
static inline void ids_append(SmallIds& row, const int32_t* b, const int32_t* e) { row.append(b, e); }

// One text -> the ids of its pre-tokens appended to `row` (no bos/eos), stopping once `budget` ids are there (truncation keeps
// a prefix: later pieces cannot matter).  false: the text is outside what this library restates (see emcid_host.h).
// Caller holds m->lock.
template <class Row>
static bool encode_text(emcid_bpe* m, const char* s, size_t len, std::string& low, Row& row, int32_t budget,
                        bool use_cache = true) {
    low.assign(s, len);
    for (size_t p = 0; p < len; ++p) {
        const unsigned char c = (unsigned char)low[p];
        if (c >= 0x80 || !servable(c) || (c == '<' && p + 1 < len && low[p + 1] == '|')) return false;
        if (c >= 'A' && c <= 'Z') low[p] = (char)(c + 32);
    }
    size_t p = 0;
    while (p < len) {
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
        if (use_cache) {
            const std::vector<int32_t>* w = nullptr;
            if (!m->word(low.data() + p, q - p, &w)) return false;
            ids_append(row, w->data(), w->data() + w->size());
        } else {        // never-seen names, worker threads: the model is only read, nothing is memoised
            static thread_local std::vector<int32_t> w;          // (one scratch block per thread, not one per pre-token)
            if (!m->merge_word(low.data() + p, q - p, w)) return false;
            ids_append(row, w.data(), w.data() + w.size());
        }
        p = q;
        if ((int32_t)row.size() >= budget) break;
    }
    return true;
}

extern "C" {

static inline void write_row(int64_t* out, const std::vector<int32_t>& row, int32_t bos, int32_t eos, int32_t max_len,
                             int32_t* length) {
    const int32_t budget = max_len - 2;
    const int32_t keep = (int32_t)row.size() < budget ? (int32_t)row.size() : budget;
    out[0] = bos;
    for (int32_t j = 0; j < keep; ++j) out[1 + j] = row[j];
    out[1 + keep] = eos;
    *length = keep + 2;
}

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
        row.clear();
        if (!encode_text(m, text + off[i], (size_t)(off[i + 1] - off[i]), low, row, max_len - 2)) {
            fallback[i] = 1;
            ++n_fallback;
            continue;
        }
        write_row(out, row, bos, eos, max_len, lengths + i);
    }
    return n_fallback;
}

int64_t emcid_bpe_encode_templated(emcid_bpe* m, const char* pre, const int64_t* pre_off, const char* suf, const int64_t* suf_off,
                                   int64_t n_templates, const char* names, const int64_t* name_off, int64_t n_names,
                                   const int32_t* tmpl_idx, const int32_t* name_idx, int64_t n, int32_t bos, int32_t eos,
                                   int32_t pad, int32_t max_len, int64_t* ids, int32_t* lengths, uint8_t* fallback,
                                   int32_t* name_last) {
    if (!m || !pre || !pre_off || !suf || !suf_off || n_templates <= 0 || !names || !name_off || n_names <= 0 || !tmpl_idx ||
        !name_idx || n < 0 || max_len < 2 || !ids || !lengths || !fallback) {
        g_error = "emcid_bpe_encode_templated: bad argument";
        return -1;
    }
    std::lock_guard<std::mutex> guard(m->lock);
    const int32_t budget = max_len - 2;
    std::string low;
    // every distinct piece is encoded once: 0 = not yet, 1 = ids there, 2 = outside the library
    struct Piece { SmallIds ids; uint8_t state = 0; };
    std::vector<Piece> P((size_t)n_templates), Sx((size_t)n_templates), Nm((size_t)n_names);
    // (template pieces come back call after call: through the model's pre-token cache; the names of a mass edit do not — a
    //  replayed set costs the same again — and go straight to the merges: no string key, no node allocation, no growing cache)
    const bool cache_names = n_names < 64;
    auto piece = [&](Piece& pc, const char* blob, const int64_t* off, int64_t k, bool use_cache = true) -> bool {
        if (pc.state == 0) pc.state = encode_text(m, blob + off[k], (size_t)(off[k + 1] - off[k]), low, pc.ids, budget, use_cache) ? 1 : 2;
        return pc.state == 1;
    };
    auto joins = [&](const char* a, const int64_t* aoff, int64_t ka, const char* b, const int64_t* boff, int64_t kb) {
        // the boundary between two pieces falls between pre-tokens iff one side is empty or white space touches it
        const int64_t la = aoff[ka + 1] - aoff[ka], lb = boff[kb + 1] - boff[kb];
        return la == 0 || lb == 0 || is_space((unsigned char)a[aoff[ka + 1] - 1]) || is_space((unsigned char)b[boff[kb]]);
    };
    for (int64_t k = 0; k < n_names; ++k)
        if (name_off[k + 1] < name_off[k]) {
            g_error = "emcid_bpe_encode_templated: index or offsets out of range";
            return -1;
        }
    for (int64_t i = 0; i < n; ++i) {
        const int64_t t = tmpl_idx[i], k = name_idx[i];
        if (t < 0 || t >= n_templates || k < 0 || k >= n_names || pre_off[t + 1] < pre_off[t] || suf_off[t + 1] < suf_off[t]) {
            g_error = "emcid_bpe_encode_templated: index or offsets out of range";
            return -1;
        }
    }
    // rows [lo, hi): concatenations of encoded pieces, or the formatted text encoded as a whole.  `shared_cache`: the caller's
    // thread (pieces are encoded on demand, through the model's pre-token cache); worker threads find every piece encoded
    // already and encode whole texts without the cache (the model is only read).
    auto assemble = [&](int64_t lo, int64_t hi, bool shared_cache, std::string& low_t) -> int64_t {
        int64_t n_fb = 0;
        std::vector<int32_t> row;
        std::string whole;
        for (int64_t i = lo; i < hi; ++i) {
            int64_t* out = ids + i * (int64_t)max_len;
            for (int32_t j = 0; j < max_len; ++j) out[j] = pad;
            lengths[i] = 0;
            fallback[i] = 0;
            if (name_last) name_last[i] = -1;
            const int64_t t = tmpl_idx[i], k = name_idx[i];
            row.clear();
            bool ok;
            if (name_off[k + 1] > name_off[k] && joins(pre, pre_off, t, names, name_off, k) && joins(names, name_off, k, suf, suf_off, t)) {
                if (shared_cache)
                    ok = piece(P[(size_t)t], pre, pre_off, t) && piece(Nm[(size_t)k], names, name_off, k, cache_names) && piece(Sx[(size_t)t], suf, suf_off, t);
                else
                    ok = P[(size_t)t].state == 1 && Nm[(size_t)k].state == 1 && Sx[(size_t)t].state == 1;
                if (ok) {
                    row.assign(P[(size_t)t].ids.begin(), P[(size_t)t].ids.end());
                    row.insert(row.end(), Nm[(size_t)k].ids.begin(), Nm[(size_t)k].ids.end());
                    // position (BOS included) of the name's last token, when the row is not cut by the length budget
                    const size_t upto = row.size();
                    row.insert(row.end(), Sx[(size_t)t].ids.begin(), Sx[(size_t)t].ids.end());
                    if (name_last && !Nm[(size_t)k].ids.empty() && row.size() <= (size_t)budget) name_last[i] = (int32_t)upto;
                }
            } else {                                   // pre-tokens may span a boundary: encode the formatted text as a whole
                whole.assign(pre + pre_off[t], (size_t)(pre_off[t + 1] - pre_off[t]));
                whole.append(names + name_off[k], (size_t)(name_off[k + 1] - name_off[k]));
                whole.append(suf + suf_off[t], (size_t)(suf_off[t + 1] - suf_off[t]));
                ok = encode_text(m, whole.data(), whole.size(), low_t, row, budget, shared_cache);
            }
            if (!ok) {
                fallback[i] = 1;
                ++n_fb;
                continue;
            }
            write_row(out, row, bos, eos, max_len, lengths + i);
        }
        return n_fb;
    };
    // A mass edit brings ~1 000 names the cache has never seen (~1 us of merges each): a few threads encode the distinct names
    // (without the cache: names of a request set are not expected back; a replayed set costs the same again), meet, and then
    // assemble the rows, each its share.
    const unsigned hw = std::thread::hardware_concurrency();
    static const int max_threads = [] { const char* e = getenv("EMCID_TOK_THREADS"); const int v = e ? atoi(e) : 4; return v < 1 ? 1 : (v > 32 ? 32 : v); }();
    // (round 6: with the flat merge table a name costs ~0.2 us; starting and joining a thread ~40: one thread up to 4 096 names)
    const int nt = n_names >= 4096 ? (int)std::min<int64_t>(std::min<int64_t>(max_threads, hw ? hw : 1), n_names / 2048) : 1;
    if (nt <= 1) return assemble(0, n, true, low);
    for (int64_t t = 0; t < n_templates; ++t) {          // the few template pieces: here, through the cache
        piece(P[(size_t)t], pre, pre_off, t);
        piece(Sx[(size_t)t], suf, suf_off, t);
    }
    std::atomic<int> arrived{0};
    std::vector<int64_t> n_fb((size_t)nt, 0);
    // No exception leaves a worker (std::terminate would take the whole Python process down) and every worker reaches the
    // rendezvous: a failed allocation marks that thread's names "not encoded" and its rows "fallback" (the caller re-encodes
    // those through the public tokenizer).  `from`: the first share this thread has to do (threads that could not be started
    // have theirs done by the caller's thread afterwards, without a rendezvous: every name is encoded or marked by then).
    auto encode_share = [&](int w, std::string& low_t) noexcept {
        int64_t k = n_names * w / nt;
        const int64_t k_hi = n_names * (w + 1) / nt;
        try {
            for (; k < k_hi; ++k) {
                Piece& pc = Nm[(size_t)k];
                pc.state = encode_text(m, names + name_off[k], (size_t)(name_off[k + 1] - name_off[k]), low_t, pc.ids, budget, false) ? 1 : 2;
            }
        } catch (...) {
            for (; k < k_hi; ++k) Nm[(size_t)k].state = 2;
        }
    };
    auto assemble_share = [&](int w, std::string& low_t) noexcept {
        const int64_t lo = n * w / nt, hi = n * (w + 1) / nt;
        try {
            n_fb[(size_t)w] = assemble(lo, hi, false, low_t);
        } catch (...) {
            for (int64_t i = lo; i < hi; ++i) fallback[i] = 1;
            n_fb[(size_t)w] = hi - lo;
        }
    };
    auto work = [&](int w) noexcept {
        std::string low_t;
        encode_share(w, low_t);
        arrived.fetch_add(1, std::memory_order_acq_rel);
        while (arrived.load(std::memory_order_acquire) < nt) std::this_thread::yield();
        assemble_share(w, low_t);
    };
    std::vector<std::thread> pool;
    int started = 1;
    try {
        pool.reserve((size_t)nt - 1);
        for (; started < nt; ++started) pool.emplace_back(work, started);
    } catch (...) {
    }
    {
        std::string low_t;
        for (int w = started; w < nt; ++w) {          // shares of threads that could not be started: their names first
            encode_share(w, low_t);
            arrived.fetch_add(1, std::memory_order_acq_rel);
        }
    }
    work(0);
    for (auto& th : pool) th.join();
    {
        std::string low_t;
        for (int w = started; w < nt; ++w) assemble_share(w, low_t);
    }
    int64_t n_fallback = 0;
    for (int64_t v : n_fb) n_fallback += v;
    return n_fallback;
}

int64_t emcid_find_token_ranges_idx(const int64_t* ids, int64_t n, int64_t S, const char* piece_ns, const int64_t* piece_off,
                                    const int32_t* piece_len, int64_t n_pieces, const char* subj, const int64_t* subj_off,
                                    const int32_t* subj_idx, int64_t n_subj, int normalize, const char* forbid, int32_t* first,
                                    int32_t* last, uint8_t* status);

int64_t emcid_find_token_ranges(const int64_t* ids, int64_t n, int64_t S, const char* piece_ns, const int64_t* piece_off,
                                const int32_t* piece_len, int64_t n_pieces, const char* subj, const int64_t* subj_off,
                                const char* forbid, int32_t* first, int32_t* last, uint8_t* status) {
    return emcid_find_token_ranges_idx(ids, n, S, piece_ns, piece_off, piece_len, n_pieces, subj, subj_off, nullptr, n, 0, forbid,
                                       first, last, status);
}

int64_t emcid_find_token_ranges_idx(const int64_t* ids, int64_t n, int64_t S, const char* piece_ns, const int64_t* piece_off,
                                    const int32_t* piece_len, int64_t n_pieces, const char* subj, const int64_t* subj_off,
                                    const int32_t* subj_idx, int64_t n_subj, int normalize, const char* forbid, int32_t* first,
                                    int32_t* last, uint8_t* status) {
    if (!ids || n < 0 || S <= 0 || !piece_ns || !piece_off || !piece_len || n_pieces <= 0 || !subj || !subj_off || !first ||
        !last || !status || n_subj < 0) {
        g_error = "emcid_find_token_ranges: bad argument";
        return -1;
    }
    if (subj_idx)
        for (int64_t i = 0; i < n; ++i)
            if (subj_idx[i] < 0 || subj_idx[i] >= n_subj) {
                g_error = "emcid_find_token_ranges: subject index out of range";
                return -1;
            }
    // normalize: the subjects arrive as the caller's raw strings; what the reference's walk searches for is
    // `sub.replace(" ", "").lower()` (causal_trace.py:1066); "[CLS]" / "[EOS]" / "" / " " are its special cases and anything
    // outside ASCII is left to the scalar walk — all of those become empty here, i.e. status 1
    std::string norm_blob;
    std::vector<int64_t> norm_off;
    if (normalize) {
        norm_off.assign((size_t)n_subj + 1, 0);
        for (int64_t k = 0; k < n_subj; ++k) {
            const char* b = subj + subj_off[k];
            const int64_t len = subj_off[k + 1] - subj_off[k];
            const size_t start = norm_blob.size();
            bool keep = len > 0 && !(len == 5 && (!std::memcmp(b, "[CLS]", 5) || !std::memcmp(b, "[EOS]", 5)));
            for (int64_t j = 0; j < len && keep; ++j) {
                const unsigned char c = (unsigned char)b[j];
                if (c >= 0x80) keep = false;
                else if (c != ' ') norm_blob.push_back((char)((c >= 'A' && c <= 'Z') ? c + 32 : c));
            }
            if (!keep) norm_blob.resize(start);
            norm_off[(size_t)k + 1] = (int64_t)norm_blob.size();
        }
        subj = norm_blob.data();
        subj_off = norm_off.data();
    }
    int64_t n_scalar = 0;
    std::string whole;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t* row = ids + i * S;
        first[i] = last[i] = 0;
        status[i] = 1;
        const int64_t si = subj_idx ? subj_idx[i] : i;
        const size_t sl = (size_t)(subj_off[si + 1] - subj_off[si]);
        bool ok = sl > 0 && subj_off[si + 1] >= subj_off[si];
        whole.clear();
        for (int64_t j = 0; j < S && ok; ++j) {
            const int64_t t = row[j];
            if (t < 0 || t >= n_pieces || piece_len[t] < 0) ok = false;
            else whole.append(piece_ns + piece_off[t], (size_t)(piece_off[t + 1] - piece_off[t]));
        }
        if (ok && forbid && *forbid && whole.find(forbid) != std::string::npos) ok = false;
        if (ok) {
            const size_t at = whole.find(subj + subj_off[si], 0, sl);
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

// ---- prefix trie of a tokenized prompt batch ------------------------------------------------------------------------------------
// Same construction and the same node numbering as emcid_amd/clip_forward.py build_trie (level by level; nodes numbered by depth,
// then by (parent, token)): the forward's row order, and with it every GEMM's operand layout, does not depend on which of the
// two built the trie.
struct emcid_trie {
    int64_t n = 0, U = 0, n_real = 0, R = 0, R_pad = 0;
    int32_t dmax = 0;
    std::vector<int64_t> token, lookup_node, inverse;
    std::vector<int32_t> depth, parent, q_rows;
    std::vector<int64_t> level_begin;      // first node of each level (dmax + 1 entries)
};

emcid_trie* emcid_trie_build(const int64_t* ids, int64_t n, int64_t S, const int64_t* lookup, int64_t bucket) {
    if (!ids || !lookup || n <= 0 || S <= 0 || bucket < 1) {
        g_error = "emcid_trie_build: bad argument";
        return nullptr;
    }
    int64_t lmax = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (lookup[i] < 0 || lookup[i] >= S) {
            g_error = "emcid_trie_build: lookup index outside the row";
            return nullptr;
        }
        if (lookup[i] > lmax) lmax = lookup[i];
    }
    auto* t = new emcid_trie();
    t->n = n;
    t->dmax = (int32_t)(lmax + 1);
    t->lookup_node.assign((size_t)n, -1);
    t->level_begin.push_back(0);
    // keys (parent + 1) * vocab + token, sorted per level by an LSD radix sort (11-bit digits): a comparison sort of the ~n
    // entries of every level was most of the build's time
    int64_t vocab = 1;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t p = 0; p <= lookup[i]; ++p) {
            const int64_t tk = ids[i * S + p];
            if (tk < 0) {
                delete t;
                g_error = "emcid_trie_build: negative token id";
                return nullptr;
            }
            if (tk >= vocab) vocab = tk + 1;
        }
    struct Entry { uint64_t key; int64_t row; };
    std::vector<Entry> cur, tmp;
    cur.reserve((size_t)n);
    tmp.reserve((size_t)n);
    std::vector<int64_t>& node_of = t->lookup_node;     // node of each prompt at the level just built (its last alive level stays)
    int64_t total = 0;
    for (int32_t p = 0; p < t->dmax; ++p) {
        cur.clear();
        uint64_t kmax = 0;
        for (int64_t i = 0; i < n; ++i)
            if (lookup[i] >= p) {
                const uint64_t key = (uint64_t)(node_of[(size_t)i] + 1) * (uint64_t)vocab + (uint64_t)ids[i * S + p];
                if (key > kmax) kmax = key;
                cur.push_back({key, i});
            }
        for (int shift = 0; shift < 64 && (kmax >> shift) != 0; shift += 11) {
            size_t count[2049] = {0};
            for (const Entry& e : cur) ++count[((e.key >> shift) & 2047) + 1];
            for (int b = 0; b < 2048; ++b) count[b + 1] += count[b];
            tmp.resize(cur.size());
            for (const Entry& e : cur) tmp[count[(e.key >> shift) & 2047]++] = e;
            cur.swap(tmp);
        }
        int64_t id = total - 1;
        for (size_t k = 0; k < cur.size(); ++k) {
            if (k == 0 || cur[k].key != cur[k - 1].key) {
                ++id;
                t->token.push_back((int64_t)(cur[k].key % (uint64_t)vocab));
                t->parent.push_back((int32_t)((int64_t)(cur[k].key / (uint64_t)vocab) - 1));
                t->depth.push_back(p);
            }
            node_of[(size_t)cur[k].row] = id;
        }
        total = id + 1;
        t->level_begin.push_back(total);
    }
    t->n_real = total;
    const int64_t pad = bucket > 1 ? (bucket - total % bucket) % bucket : 0;
    t->U = total + pad;
    for (int64_t k = 0; k < pad; ++k) {       // padding nodes: copies of the first root token at depth 0, attending to themselves
        t->token.push_back(t->token[0]);
        t->depth.push_back(0);
        t->parent.push_back(-1);
    }
    // distinct lookup nodes, sorted, and each prompt's index among them: mark the nodes, then one scan in node order
    std::vector<int32_t> rank((size_t)total, -1);
    for (int64_t i = 0; i < n; ++i) rank[(size_t)t->lookup_node[(size_t)i]] = 0;
    t->q_rows.clear();
    for (int64_t u = 0; u < total; ++u)
        if (rank[(size_t)u] == 0) {
            rank[(size_t)u] = (int32_t)t->q_rows.size();
            t->q_rows.push_back((int32_t)u);
        }
    t->R = (int64_t)t->q_rows.size();
    t->inverse.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) t->inverse[(size_t)i] = rank[(size_t)t->lookup_node[(size_t)i]];
    if (bucket > 1 && t->R % bucket) {
        const int32_t q0 = t->q_rows[0];
        t->q_rows.resize((size_t)(t->R + (bucket - t->R % bucket)), q0);
    }
    t->R_pad = (int64_t)t->q_rows.size();
    return t;
}

void emcid_trie_sizes(const emcid_trie* t, int64_t* U, int64_t* n_real, int64_t* dmax, int64_t* R_pad) {
    if (!t) return;
    if (U) *U = t->U;
    if (n_real) *n_real = t->n_real;
    if (dmax) *dmax = t->dmax;
    if (R_pad) *R_pad = t->R_pad;
}

int64_t emcid_trie_packed_bytes(const emcid_trie* t) {
    if (!t) return -1;
    const int64_t i32s = t->U + t->R_pad + t->U * t->dmax;
    return 8 * (t->U + 2 * t->n) + 4 * (i32s + (i32s & 1));
}

int emcid_trie_export(const emcid_trie* t, void* out, int64_t out_bytes) {
    if (!t || !out || out_bytes < emcid_trie_packed_bytes(t)) {
        g_error = "emcid_trie_export: bad argument";
        return -1;
    }
    int64_t* p64 = (int64_t*)out;
    std::memcpy(p64, t->token.data(), (size_t)t->U * 8);
    std::memcpy(p64 + t->U, t->lookup_node.data(), (size_t)t->n * 8);
    std::memcpy(p64 + t->U + t->n, t->inverse.data(), (size_t)t->n * 8);
    int32_t* p32 = (int32_t*)(p64 + t->U + 2 * t->n);
    std::memcpy(p32, t->depth.data(), (size_t)t->U * 4);
    std::memcpy(p32 + t->U, t->q_rows.data(), (size_t)t->R_pad * 4);
    int32_t* anc = p32 + t->U + t->R_pad;
    const int32_t D = t->dmax;
    std::memset(anc, 0, (size_t)t->U * D * 4);
    for (int64_t u = 0; u < t->U; ++u) {            // parents precede their children: copy the parent's chain, append self
        const int32_t d = t->depth[(size_t)u], par = t->parent[(size_t)u];
        if (d > 0) std::memcpy(anc + u * D, anc + (int64_t)par * D, (size_t)d * 4);
        anc[u * D + d] = (int32_t)u;
    }
    return 0;
}

void emcid_trie_destroy(emcid_trie* t) { delete t; }

}  // extern "C"
