/*
 * szl_oracle.h — CPU restatement (plain C) of SharpZipLib v1.4.2's Zip.Compression hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it, and
 * only as the checker / the timed CPU baseline.  The shipped library (sharpziplib_amd/csrc)
 * never links or calls this code.
 *
 * PARITY STATUS: "parity unpinned" for the ENCODER bits — the reference's own tests hold no
 * expected compressed bytes for Deflater and no .NET runtime exists in
 * this image to run the managed code.  The encoder is pinned by (i) being a line-by-line
 * restatement of the files cited at every function, (ii) the tiny vectors of SURVEY.md App. C.8
 * incl. the reference's own fixture payload T/Zip/ZipCorruptionHandling.cs:52-54
 * ("testfile contents\n" -> 2b492d2e49cbcc495548cecf2b49cd2b29e60200), (iii) every output
 * being decodable by system zlib.  PINNED by reference known-answer tests: CRC32/Adler32
 * (T/Checksum/ChecksumTests.cs:31,86,95,104,114,127,136,145) and the Inflater fixtures
 * (T/Zip/ZipCorruptionHandling.cs:12-16 must throw, :52-54 must inflate).
 *
 * Reference paths below are relative to /root/reference/src/ICSharpCode.SharpZipLib/ :
 *   C/  = Zip/Compression/        CS/ = Zip/Compression/Streams/      K/ = Checksum/
 */
#ifndef SZL_ORACLE_H
#define SZL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes returned (negative) by the inflater; mirror the exceptions of C/Inflater.cs ---- */
enum {
    SZO_OK = 0,
    SZO_ERR_HEADER_CHECKSUM = -1,   /* "Header checksum illegal"        C/Inflater.cs:224 */
    SZO_ERR_METHOD_UNKNOWN = -2,    /* "Compression Method unknown"     C/Inflater.cs:229 */
    SZO_ERR_ILLEGAL_LEN_CODE = -3,  /* "Illegal rep length code"        C/Inflater.cs:325 */
    SZO_ERR_ILLEGAL_DIST_CODE = -4, /* "Illegal rep dist code"          C/Inflater.cs:358 */
    SZO_ERR_ADLER_MISMATCH = -5,    /* "Adler chksum doesn't match"     C/Inflater.cs:413 */
    SZO_ERR_UNKNOWN_BLOCK = -6,     /* "Unknown block type"             C/Inflater.cs:486 */
    SZO_ERR_BROKEN_STORED = -7,     /* "broken uncompressed block"      C/Inflater.cs:511 */
    SZO_ERR_CODELEN_ZERO = -8,      /* "Encountered invalid codelength 0" C/InflaterHuffmanTree.cs:191-193 */
    SZO_ERR_DYN_HEADER = -9,        /* ValueOutOfRange / StreamDecodingException C/InflaterDynHeader.cs:50-52,83,106,114 */
    SZO_ERR_WINDOW_FULL = -10,      /* "Window full" CS/OutputWindow.cs:37,66 */
    SZO_ERR_STATE = -11,            /* InvalidOperationException family */
    SZO_ERR_ARG = -12,              /* ArgumentOutOfRange family */
    SZO_ERR_INDEX_RANGE = -13       /* IndexOutOfRangeException out of InflaterHuffmanTree.BuildTree (C/InflaterHuffmanTree.cs:133-166):
                                       over-subscribed code lengths push a canonical code to >= 65536 and
                                       DeflaterHuffman.BitReverse (C/DeflaterHuffman.cs:924-930) indexes bit4Reverse[16+] */
};

/* ------------------------------------------------------------------ checksums (K/) */
uint32_t szo_crc32(uint32_t crc_value, const uint8_t *p, size_t n);   /* crc_value = current Crc32.Value (0 initially) */
uint32_t szo_adler32(uint32_t adler_value, const uint8_t *p, size_t n); /* adler_value = current Adler32.Value (1 initially) */

/* ------------------------------------------------------------------ Deflater (C/Deflater.cs) */
typedef struct szo_deflater szo_deflater;

/* Optional trace of the token stream / block decisions (used to diff GPU intermediates). */
typedef struct szo_block_info {
    int64_t first_token;   /* global token index of first token in the block */
    int32_t ntokens;
    int32_t type;          /* 0 stored, 1 static, 2 dynamic */
    int32_t last;          /* BFINAL */
    int32_t stored_offset; /* window index handed to FlushBlock (may be negative) */
    int32_t stored_len;
    int32_t opt_len;       /* bits (after the static/dyn min) */
    int32_t static_len;
    int64_t bit_start;     /* absolute bit offset of the block header in the raw deflate stream */
} szo_block_info;

typedef struct szo_trace {
    /* tokens: lit => dist=0, val=literal ; match => dist>0, val=len */
    uint32_t *tok;        /* packed: dist<<16 | val (val = literal byte, or len) */
    size_t tok_cap, tok_n;
    szo_block_info *blk;
    size_t blk_cap, blk_n;
} szo_trace;

szo_deflater *szo_deflater_new(int level, int no_zlib_header_or_footer); /* level -1..9; NULL on bad level */
void szo_deflater_free(szo_deflater *d);
void szo_deflater_reset(szo_deflater *d);
int  szo_deflater_set_level(szo_deflater *d, int level);
void szo_deflater_set_strategy(szo_deflater *d, int strategy);         /* 0 Default, 1 Filtered, 2 HuffmanOnly */
int  szo_deflater_set_dictionary(szo_deflater *d, const uint8_t *p, int n);
int  szo_deflater_set_input(szo_deflater *d, const uint8_t *p, int n);  /* borrows p until needs_input */
void szo_deflater_flush(szo_deflater *d);
void szo_deflater_finish(szo_deflater *d);
int  szo_deflater_deflate(szo_deflater *d, uint8_t *out, int len);
int  szo_deflater_needs_input(const szo_deflater *d);
int  szo_deflater_is_finished(const szo_deflater *d);
int64_t szo_deflater_total_in(const szo_deflater *d);
int64_t szo_deflater_total_out(const szo_deflater *d);
uint32_t szo_deflater_adler(const szo_deflater *d);
void szo_deflater_set_trace(szo_deflater *d, szo_trace *t);

/* One-shot helper: SetInput(all) [chunked to <2^30 per call]; optional Flush; Finish; drain.
 * Returns compressed size or negative on overflow of out_cap. */
int64_t szo_deflate_oneshot(const uint8_t *in, size_t n, int level, int nowrap, int strategy,
                            int flush_before_finish, uint8_t *out, size_t out_cap, szo_trace *trace);

/* ------------------------------------------------------------------ Inflater (C/Inflater.cs) */
typedef struct szo_inflater szo_inflater;
szo_inflater *szo_inflater_new(int no_header);
void szo_inflater_free(szo_inflater *s);
void szo_inflater_reset(szo_inflater *s);
int  szo_inflater_set_input(szo_inflater *s, const uint8_t *p, int n);
int  szo_inflater_set_dictionary(szo_inflater *s, const uint8_t *p, int n);
int  szo_inflater_inflate(szo_inflater *s, uint8_t *out, int count);  /* >=0 bytes, <0 error */
int  szo_inflater_needs_input(const szo_inflater *s);
int  szo_inflater_needs_dictionary(const szo_inflater *s);
int  szo_inflater_is_finished(const szo_inflater *s);
int  szo_inflater_remaining_input(const szo_inflater *s);
int64_t szo_inflater_total_in(const szo_inflater *s);
int64_t szo_inflater_total_out(const szo_inflater *s);
uint32_t szo_inflater_adler(const szo_inflater *s);

/* One-shot: inflate `in` fully into out; returns bytes or negative error; *consumed = TotalIn. */
int64_t szo_inflate_oneshot(const uint8_t *in, size_t n, int no_header, uint8_t *out, size_t out_cap,
                            size_t *consumed);
/* Test aid: number of incomplete code-length sets with codes of 10+ bits built since the last reset (see iht_build). */
/* bench infrastructure (szl_parallel.c): n_slices independent one-shot raw Deflaters on `threads` host threads */
int64_t szo_deflate_slices_mt(const uint8_t *in, size_t slice_len, int n_slices, int level, int threads, uint64_t *out_lens);
int szo_quirk_sets_seen(int reset);
int szo_tree_lengths(const int16_t *freqs, int nsyms, int minCodes, int maxLen, uint8_t *len_out, int *numCodes_out);   /* test tap: Tree.BuildTree + BuildLength */
/* test hooks: BuildTree's table (returns treeSize or an error) and one GetSymbol with `avail` bits of input left */
int szo_iht_table(const uint8_t *codeLengths, int n, int16_t *out, int cap);
int szo_iht_symbol(const int16_t *tree, int treeSize, uint32_t bits, int avail, int *dropped);
int szo_sm_script(const uint8_t *buf, int n, const int32_t *ops, int nops, const int16_t *tree, int treeSize, int32_t *results);
/* Same, asking for one byte per Inflate() call: *produced = bytes delivered before the error (or in total). */
int64_t szo_inflate_probe(const uint8_t *in, size_t n, int no_header, uint8_t *out, size_t out_cap, size_t *consumed,
                          size_t *produced);

#ifdef __cplusplus
}
#endif
#endif
