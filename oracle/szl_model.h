/*
 * szl_model.h — CPU model of the PARALLEL decomposition used by the HIP kernels
 * (sharpziplib_amd/csrc).  TEST INFRASTRUCTURE ONLY, like the rest of oracle/.
 *
 * The reference's DeflateSlow (C/DeflaterEngine.cs:741-855) is restated here as
 *   stage A  hash links        link[q]  = distance to the previous inserted position with the same hash
 *   stage B  match tables      M2[p],Mq[p] = FindLongestMatch(p) from matchLen=2 with full / quarter chain budget
 *   stage C  parse             orbit of 0 under J(p) = next "clean" (matchLen==2) iteration position
 * which is what makes a single deflate stream data-parallel.
 * tests/ check that this model's token stream equals the oracle engine's token trace on every
 * input class; the GPU kernels are then diffed stage-by-stage against the model.
 */
#ifndef SZL_MODEL_H
#define SZL_MODEL_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* A "segment" is the span of input between two Flush()/Finish() calls of one stream.
 * seg_start..seg_end are absolute positions inside the stream buffer d[0..seg_end). */
typedef struct szm_params {
    int good, nice, max_chain; /* DeflaterConstants.cs:124-139 for the level */
    int strategy;              /* 0 Default, 1 Filtered, 2 HuffmanOnly */
} szm_params;

int szm_level_params(int level, szm_params *out); /* returns 0, or -1 if level is not a DEFLATE_SLOW level (5..9) */

/* Stage A. link[q] for q in [0,n): 0 = none / farther than 32767 / q not inserted.
 * seg_ends[0..nseg) ascending, last == n: position q is inserted iff seg_end(q) - q >= 3. */
void szm_links(const uint8_t *d, size_t n, const size_t *seg_ends, size_t nseg, uint16_t *link);

/* Stage B for positions [seg_start, seg_end). Entry = len | dist<<16, 0 = no match (>=3). */
void szm_match_tables(const uint8_t *d, size_t seg_start, size_t seg_end, const uint16_t *link,
                      const szm_params *P, uint32_t *m2, uint32_t *mq);

/* Stage C, sequential orbit. Tokens: literal = byte ; match = dist<<16 | len.
 * Returns number of tokens written (tok must hold seg_end-seg_start entries).
 * stats[0] += fallback walks (prevLen >= nice'), stats[1] += nodes. */
size_t szm_parse(const uint8_t *d, size_t seg_start, size_t seg_end, const uint16_t *link,
                 const uint32_t *m2, const uint32_t *mq, const szm_params *P, uint32_t *tok, uint64_t *stats);

/* Stage C, range-speculative form (mirrors kernels C1..C3): ranges of R positions are walked
 * from a fresh state, fixed up from the predecessor's exit, and stitched.  Must produce the
 * same tokens as szm_parse.  stats[0] += ranges that did not merge. */
size_t szm_parse_ranges(const uint8_t *d, size_t seg_start, size_t seg_end, const uint16_t *link,
                        const uint32_t *m2, const uint32_t *mq, const szm_params *P, size_t R,
                        uint32_t *tok, uint64_t *stats);

/* Block table for a segment's token stream (DeflaterEngine.cs:841-852, :750-768):
 * fills first_token[]/ntok[]/last[] ; returns number of blocks. `finish` = segment ended by Finish(). */
/* chain compression: links between positions that share four bytes + the hash-chain hops each one passes; the
 * compressed walk must give the tables of szm_match_tables.  steps (optional) counts the candidates it examines. */
void szm_links4(const uint8_t *d, size_t n, const uint16_t *link, uint16_t *link4, uint16_t *skip4);
void szm_match_tables_c4(const uint8_t *d, size_t n, size_t seg_start, size_t seg_end, const uint16_t *link, const uint16_t *link4,
                         const uint16_t *skip4, const szm_params *P, uint32_t *m2, uint32_t *mq, uint64_t *steps);
/* the compressed walk in the shape of the device kernel (hop counts saturating at 255, slow routine while best_len is 2) */
void szm_match_tables_k6(const uint8_t *d, size_t n, size_t seg_start, size_t seg_end, const uint16_t *link, const uint16_t *link4,
                         const uint16_t *skip4, const szm_params *P, uint32_t *m2, uint32_t *mq, uint64_t *steps);
/* the same without the slow routine: the first candidate is the first chain element with the same three bytes (device form) */
void szm_links4e(const uint8_t *d, size_t n, const uint16_t *link, uint16_t *link4, uint8_t *skip8, uint16_t *e3d, uint8_t *e3h, int dist_cap);
void szm_match_tables_k7(const uint8_t *d, size_t n, size_t seg_start, size_t seg_end, const uint16_t *link, int dist_cap,
                         const szm_params *P, uint32_t *m2, uint32_t *mq, uint64_t *steps);
/* first clean iteration >= at_least of the parse that starts clean at `from` */
size_t szm_first_node(const uint8_t *d, size_t seg_end, const uint16_t *link, const uint32_t *m2, const uint32_t *mq,
                      const szm_params *P, size_t from, size_t at_least);
size_t szm_block_table(const uint32_t *tok, size_t ntok, int finish, int64_t *first_token, int32_t *count, int32_t *last);

/* base_of(s): window base in effect for an iteration starting at absolute position s (App. A.2). */
int64_t szm_base_of(int64_t s);

/* Analysis helper: marks the positions at which the parse reads the match tables; returns how many. */
size_t szm_parse_needed(const uint8_t *d, size_t seg_start, size_t seg_end, const uint16_t *link, const uint32_t *m2,
                        const uint32_t *mq, const szm_params *P, uint8_t *needed);

/* Model of k_match_lazy: positions evaluated by walkers started every `stride` positions inside tiles of `tile` positions. */
size_t szm_lazy_eval_set(const uint8_t *d, size_t seg_start, size_t seg_end, const uint16_t *link, const uint32_t *m2,
                         const uint32_t *mq, const szm_params *P, size_t tile, size_t stride, uint8_t *evaluated);

/* ---- DeflateFast (levels 1-4), see szl_model.c ---- */
typedef struct szm_fast_params { int nice, max_chain, max_lazy, strategy; } szm_fast_params;
int szm_fast_level_params(int level, szm_fast_params *out); /* 0, or -1 if level is not 1..4 */
int64_t szm_base_of_fast(int64_t s);
/* Sequential greedy parse over the flag-filtered all-positions chain. flags[q] (bytes) in: history below
 * seg_start, out: inserted flag of every position in the segment. Returns the token count. */
size_t szm_fast_parse(const uint8_t *d, size_t seg_start, size_t seg_end, const uint16_t *link,
                      const szm_fast_params *P, uint8_t *flags, uint32_t *tok);
/* Range-parallel fixpoint form (what the kernels do); *iters = iterations until (flags, exits) repeat. */
size_t szm_fast_parse_fixpoint(const uint8_t *d, size_t seg_start, size_t seg_end, const uint16_t *link,
                               const szm_fast_params *P, size_t R, uint8_t *flags, uint32_t *tok, uint64_t *iters);

#ifdef __cplusplus
}
#endif
#endif
