/* emcid_host.h — C ABI of libemcid_host.so: the host-side text work and cache reads on the edit path (no GPU, no HIP).
 *
 * The reference tokenizes every prompt of an edit with the pipeline's own Hugging Face CLIP tokenizer
 * (`tokenize_prompts`, emcid/compute_z.py:65: `tokenizer(prompts, return_tensors="pt", padding=True, truncation=True)`, called
 * from get_module_input_output_at_words, compute_z.py:2284) and finds each subject's last token by decoding the ids
 * (`find_token_range`, experiments/causal_trace.py:1057, called per prompt at compute_z.py:2287-2290).  On a 1 000-concept edit that is 3 000 prompts: 9-17 ms in the HF
 * tokenizer and 3-4 ms in the search, more than the whole GPU solve.  These two entry points restate exactly that work for
 * the common case (ASCII prompts, a CLIP byte-level BPE vocabulary) and REPORT, per prompt, when a prompt is outside that
 * case — the caller then sends just those prompts through the HF tokenizer.  `emcid_amd/host_text.py` is the binding; it
 * checks the tokenizer's configuration and a probe set against the HF tokenizer before it trusts this library.
 */
#ifndef EMCID_HOST_H
#define EMCID_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct emcid_bpe emcid_bpe;

int emcid_host_abi_version(void);

/* A CLIP byte-level BPE model (tokenizers' `BPE` with `end_of_word_suffix`, behind the CLIP normalizer / pre-tokenizer:
 * NFC, \s+ -> ' ', lowercase; split on  's|'t|'re|'ve|'m|'ll|'d|\p{L}+|\p{N}|[^\s\p{L}\p{N}]+ ; ByteLevel).
 * vocab: token i is the UTF-8 string vocab_bytes[vocab_off[i] .. vocab_off[i+1]) with id vocab_ids[i].
 * merges: n_merges (left id, right id) pairs in rank order; the merged token (left + right) must be in the vocabulary.
 * Returns NULL (and sets emcid_host_last_error) on an inconsistent model. */
emcid_bpe* emcid_bpe_create(const char* vocab_bytes, const int64_t* vocab_off, const int32_t* vocab_ids, int64_t n_vocab,
                            const int32_t* merges, int64_t n_merges, const char* end_of_word_suffix);
void emcid_bpe_destroy(emcid_bpe* m);

/* Encode n texts; text i is text[off[i] .. off[i+1]).  Row i of ids (n x max_len, int64) becomes
 * bos, tokens (at most max_len - 2), eos, then `pad` up to max_len; lengths[i] = tokens written including bos/eos.
 * fallback[i] = 1 and the row is left as pad when text i is outside what this library restates exactly: a byte outside
 * printable ASCII / ASCII white space, the sequence "<|" (special-token syntax), or a character missing from the vocabulary.
 * Returns the number of fallback rows, or -1 on a bad argument.  Thread-safe (one lock per model). */
int64_t emcid_bpe_encode_batch(emcid_bpe* m, const char* text, const int64_t* off, int64_t n, int32_t bos, int32_t eos,
                               int32_t pad, int32_t max_len, int64_t* ids, int32_t* lengths, uint8_t* fallback);

/* The token range of a subject inside a tokenized prompt, as the reference's find_token_range walks it, for n rows at once.
 * ids: n x S int64.  Per token id t < n_pieces: piece_ns[piece_off[t] .. piece_off[t+1]) = decode([t]) without spaces
 * (ASCII), piece_len[t] = len(decode([t])) WITH spaces (what the reference's walk counts); piece_len[t] < 0 marks a token
 * this table cannot serve (non-ASCII piece).  subj: the subjects, already lower-cased and stripped of spaces,
 * subj[subj_off[i] .. subj_off[i+1]).  forbid (may be NULL or empty): a string whose presence in a row's concatenated pieces
 * means the tokenizer's own decode would differ from the concatenation (CLIP's end-of-word suffix formed across tokens).
 * first[i], last[i] = the [first, last) token range; status[i] = 0 found, 1 = take the scalar path (unservable token, empty
 * subject, forbidden string, subject not found or never covered).
 * Returns the number of rows with status != 0, or -1 on a bad argument. */
int64_t emcid_find_token_ranges(const int64_t* ids, int64_t n, int64_t S, const char* piece_ns, const int64_t* piece_off,
                                const int32_t* piece_len, int64_t n_pieces, const char* subj, const int64_t* subj_off,
                                const char* forbid, int32_t* first, int32_t* last, uint8_t* status);

/* emcid_bpe_encode_batch for prompts that are `template.format(name)` with ONE "{}" per template (the mass-edit case: a few
 * templates x many names; reference compute_z.py:2278-2283 formats every prompt, :65 tokenizes every string): prompt i is
 * pre[tmpl_idx[i]] + names[name_idx[i]] + suf[tmpl_idx[i]].  Every distinct piece is encoded once and rows are concatenations
 * whenever white space (or an empty piece) separates the pieces — pre-tokens never span white space — else the formatted text is
 * encoded as a whole; same ids, lengths and fallback flags as emcid_bpe_encode_batch on the formatted strings.  name_last (may be
 * NULL): per row the position of the LAST token of the name inside the row (BOS counted) where the row is such a concatenation
 * and not cut by max_len, else -1 — where find_token_range's walk (causal_trace.py:1046-1103) will put the lookup token unless the
 * name also occurs earlier in the prompt; the caller still runs the walk, but may do so after it has started the GPU. */
int64_t emcid_bpe_encode_templated(emcid_bpe* m, const char* pre, const int64_t* pre_off, const char* suf, const int64_t* suf_off,
                                   int64_t n_templates, const char* names, const int64_t* name_off, int64_t n_names,
                                   const int32_t* tmpl_idx, const int32_t* name_idx, int64_t n, int32_t bos, int32_t eos,
                                   int32_t pad, int32_t max_len, int64_t* ids, int32_t* lengths, uint8_t* fallback,
                                   int32_t* name_last);

/* emcid_find_token_ranges with the subjects given once: row i searches subject subj_idx[i] of n_subj (subj_idx NULL: row i
 * searches subject i).  normalize != 0: the subjects are the caller's raw strings and are lower-cased and stripped of ' '
 * here, as find_token_range does (causal_trace.py:1066); its special subjects ("[CLS]", "[EOS]", "", " ") and anything
 * outside ASCII come back with status 1. */
int64_t emcid_find_token_ranges_idx(const int64_t* ids, int64_t n, int64_t S, const char* piece_ns, const int64_t* piece_off,
                                    const int32_t* piece_len, int64_t n_pieces, const char* subj, const int64_t* subj_off,
                                    const int32_t* subj_idx, int64_t n_subj, int normalize, const char* forbid, int32_t* first,
                                    int32_t* last, uint8_t* status);

/* The prefix trie of n tokenized prompts, each cut behind its lookup token (the edit's forward runs once per DISTINCT causal
 * prefix: emcid_amd/clip_forward.py).  Replaces nothing in the reference (which runs the dense batch, compute_z.py:2308); it is
 * the host-side index build of this implementation's forward, moved out of numpy.  ids: n x S int64 (>= 0), lookup[i] in [0, S).
 * Nodes are numbered level by level, inside a level by (parent node, token); node and query-row counts are padded to a multiple
 * of `bucket` (padding nodes: the first root token at depth 0, each attending to itself; padding query rows: the first one).
 * emcid_trie_export writes ONE packed image (a single host-to-device copy), 8-byte aligned, in this order:
 *   int64 token[U] | int64 lookup_node[n] | int64 lookup_in_query[n] | int32 depth[U] | int32 query_rows[R_pad] |
 *   int32 anc[U][dmax]   (ancestor chain root..node, zero padded) */
typedef struct emcid_trie emcid_trie;
emcid_trie* emcid_trie_build(const int64_t* ids, int64_t n, int64_t S, const int64_t* lookup, int64_t bucket);
void emcid_trie_sizes(const emcid_trie* t, int64_t* U, int64_t* n_real, int64_t* dmax, int64_t* R_pad);
int64_t emcid_trie_packed_bytes(const emcid_trie* t);
int emcid_trie_export(const emcid_trie* t, void* out, int64_t out_bytes);
void emcid_trie_destroy(emcid_trie* t);

/* The v* rows of an edit straight from the reference's cache files (emcid/emcid_main.py:885-899 reads them with np.load, one
 * per request; :951-968 writes them with np.savez(f, v_star=...)).  File i is paths[off[i] .. off[i+1]); each is read whole
 * and accepted when its FIRST zip member is "<member>.npy", stored uncompressed, dtype <f4 or <f8, shape (width,) or
 * (1, width); the values go to out[i * ld .. i * ld + width) as float32 (f8 rounded to nearest, like numpy's astype).
 * status[i]: 0 = row written, 1 = no such file, 2 = not such a file (the caller reads it with numpy and gets numpy's
 * behaviour, errors included).  The files are dealt to at most n_threads threads (at least 64 files each).
 * Returns the number of rows with status != 0, or -1 on a bad argument.  `out` may be page-locked memory. */
int64_t emcid_read_npz_rows_f32(const char* paths, const int64_t* off, int64_t n, const char* member, int64_t width, float* out,
                                int64_t ld, uint8_t* status, int32_t n_threads);

const char* emcid_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
