/*
 * szl.h — C ABI of libszl_amd.so: the MI355X-native DEFLATE engine behind SharpZipLib's
 * Deflater / Inflater.  This is the drop-in boundary: the reference has no FFI,
 * so the boundary is the public member set of ICSharpCode.SharpZipLib.Zip.Compression.Deflater /
 * Inflater; every entry point below names the member it replaces.  The C# shim classes
 *  P/Invoke these 1:1; DeflaterOutputStream / InflaterInputStream stay managed
 * and only see the two codec classes.
 *
 * Conventions: opaque handles; caller-owned buffers; plain pointers and sizes; no callbacks.
 * Functions returning int return >= 0 on success and a negative szl_status on error (the shim
 * maps each code to the reference's exception type + message).  A handle is single-threaded like
 * the reference objects (C/Deflater.cs:10-11); the library is re-entrant across handles.
 * There is NO CPU fallback: every entry point that compresses/decompresses runs HIP kernels on
 * gfx950 and fails with SZL_E_DEVICE if no device is usable.
 *
 * Reference paths: C/ = src/ICSharpCode.SharpZipLib/Zip/Compression/, K/ = .../Checksum/.
 */
#ifndef SZL_H
#define SZL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum szl_status {
    SZL_OK = 0,
    SZL_E_ARG = -1,            /* ArgumentOutOfRangeException / ArgumentNullException (C/Deflater.cs:184-187, C/DeflaterEngine.cs:148-176) */
    SZL_E_STATE = -2,          /* InvalidOperationException ("Finish() already called" C/Deflater.cs:333-336, "Old input was not completely processed" C/DeflaterEngine.cs:163-166, "Dictionary is not needed" C/Inflater.cs:580) */
    SZL_E_DEVICE = -3,         /* HIP error / no gfx950 device: surfaces as SharpZipBaseException */
    SZL_E_NOMEM = -4,
    SZL_E_UNSUPPORTED = -5,    /* API-legal in the reference but not reproduced here */
    SZL_E_OUTPUT_TOO_SMALL = -6,
    /* Inflater errors == the SharpZipBaseException messages of C/Inflater.cs */
    SZL_E_HEADER_CHECKSUM = -16,   /* "Header checksum illegal"            C/Inflater.cs:224 */
    SZL_E_METHOD_UNKNOWN = -17,    /* "Compression Method unknown"         C/Inflater.cs:229 */
    SZL_E_ILLEGAL_LEN_CODE = -18,  /* "Illegal rep length code"            C/Inflater.cs:325 */
    SZL_E_ILLEGAL_DIST_CODE = -19, /* "Illegal rep dist code"              C/Inflater.cs:358 */
    SZL_E_ADLER_MISMATCH = -20,    /* "Adler chksum doesn't match"         C/Inflater.cs:413 */
    SZL_E_UNKNOWN_BLOCK = -21,     /* "Unknown block type"                 C/Inflater.cs:486 */
    SZL_E_BROKEN_STORED = -22,     /* "broken uncompressed block"          C/Inflater.cs:511 */
    SZL_E_CODELEN_ZERO = -23,      /* "Encountered invalid codelength 0"   C/InflaterHuffmanTree.cs:191-193 */
    SZL_E_DYN_HEADER = -24,        /* ValueOutOfRange/StreamDecodingException C/InflaterDynHeader.cs:50-52,83,106,114 */
    SZL_E_UNEXPECTED_EOF = -25,    /* batch inflate only: input exhausted before the final block (CS/InflaterInputStream.cs:494) */
    SZL_E_WINDOW_FULL = -26,       /* CS/OutputWindow.cs:37,66 */
    SZL_E_CODE_OVERSUBSCRIBED = -27,/* over-subscribed code lengths in a dynamic header: the reference's InflaterHuffmanTree.BuildTree
                                       throws IndexOutOfRangeException out of DeflaterHuffman.BitReverse (C/InflaterHuffmanTree.cs:133-166,
                                       C/DeflaterHuffman.cs:924-930); the shim rethrows that type */
    SZL_E_INDEX = -28              /* IndexOutOfRangeException out of DeflaterEngine.UpdateHash (C/DeflaterEngine.cs:409): SetLevel from level 0 to a
                                       coded level while DeflateStored stands at one of the last two bytes of a full window reads
                                       window[strstart + 1] past the array.  A caller who drains Deflate() never gets there (the last
                                       engine call slides the window first, :371); kept as a guard, not reachable through this API */
} szl_status;

const char *szl_strerror(int status);
const char *szl_last_error(void);        /* thread-local detail string of the last failing call */
int szl_device_count(void);              /* number of usable gfx950 devices (0 => every codec call fails with SZL_E_DEVICE) */
int szl_set_device(int ordinal);         /* device used by handles created afterwards on this thread */

/* ------------------------------------------------------------------------------------------
 * Checksums on device memory or host memory (K/Crc32.cs:138, K/Adler32.cs:134).
 * `value` is the running IChecksum.Value (0 for a fresh Crc32, 1 for a fresh Adler32).
 * ------------------------------------------------------------------------------------------ */
int szl_crc32(uint32_t value, const void *host_data, size_t n, uint32_t *out);
int szl_adler32(uint32_t value, const void *host_data, size_t n, uint32_t *out);

/* ------------------------------------------------------------------------------------------
 * Deflater — replaces ICSharpCode.SharpZipLib.Zip.Compression.Deflater (C/Deflater.cs)
 * ------------------------------------------------------------------------------------------ */
typedef struct szl_deflater szl_deflater;

/* Deflater(int level, bool noZlibHeaderOrFooter) C/Deflater.cs:178 ; level -1 => 6 ; returns NULL and
 * sets szl_last_error on a bad level (ArgumentOutOfRangeException :184-187) or device failure.
 *
 * Levels and speed.  5-9 (DeflateSlow) are the accelerated path: every position's match search is independent, so one stream
 * spreads over the whole device (level 6: ~16 GiB/s per MI355X, bench.py).  0 is a copy with block headers.  1-4 (DeflateFast)
 * are COMPATIBILITY-ONLY: bit-identical, but the dictionary of DeflateFast depends on its own output (C/DeflaterEngine.cs:697-712),
 * which makes a stream a sequential recurrence — one wavefront per stream, ~2 MiB/s for a single stream (a CPU core does ~95)
 * and ~0.5 GiB/s over thousands of streams in one batch call.  Level 5 is faster than that by an order of magnitude AND
 * compresses better; pick 1-4 on the device only to reproduce bytes an existing consumer expects. */
szl_deflater *szl_deflater_create(int level, int no_zlib_header_or_footer);
void szl_deflater_destroy(szl_deflater *d);
int szl_deflater_reset(szl_deflater *d);                                   /* Reset()          C/Deflater.cs:204 */
int szl_deflater_set_level(szl_deflater *d, int level);                    /* SetLevel(int)    C/Deflater.cs:349 */
int szl_deflater_get_level(const szl_deflater *d);                         /* GetLevel()       C/Deflater.cs:371 */
int szl_deflater_set_strategy(szl_deflater *d, int strategy);              /* SetStrategy      C/Deflater.cs:385 ; 0 Default 1 Filtered 2 HuffmanOnly */
int szl_deflater_set_dictionary(szl_deflater *d, const uint8_t *p, int n); /* SetDictionary    C/Deflater.cs:559 */
/* SetInput(byte[],int,int) C/Deflater.cs:331.  The reference borrows the caller's array until
 * IsNeedingInput; this library COPIES the bytes into its staging buffer before returning, so the
 * caller may reuse the array immediately.  As in the reference, IsNeedingInput is false until the next
 * Deflate() call has "consumed" the input (it returns 0 before Flush/Finish), and a second SetInput before
 * that fails with SZL_E_STATE ("Old input was not completely processed", C/DeflaterEngine.cs:163-166). */
int szl_deflater_set_input(szl_deflater *d, const uint8_t *p, int n);
int szl_deflater_flush(szl_deflater *d);                                   /* Flush()          C/Deflater.cs:252 */
int szl_deflater_finish(szl_deflater *d);                                  /* Finish()         C/Deflater.cs:262 */
/* Deflate(byte[],int,int) C/Deflater.cs:427 : returns bytes written into out[0..len).  Legal returns
 * of 0 with IsNeedingInput==true happen before Flush/Finish (the whole pending segment is compressed on
 * the device inside the first Deflate call after Flush()/Finish(), then drained across calls). */
int szl_deflater_deflate(szl_deflater *d, uint8_t *out, int len);
/* The Deflate() loop of DeflaterOutputStream (CS/DeflaterOutputStream.cs:100-118 Finish, :242-272 Deflate) without its copies: ALL the
 * bytes the next Deflate() calls would hand out, in place — *p points into the object's pinned output queue (filled by DMA), *n is their
 * count (0: nothing to hand out now, exactly where Deflate() returns 0).  They count as handed out (TotalOut, IsFinished) and stay
 * readable until the next call on the object.  A device-aware DeflaterOutputStream (sharpziplib_amd/dotnet/DeflaterOutputStream.Device.cs,
 * streams.py) writes them to its base stream in one Write instead of buffer_.Length bytes at a time — 512 by default (:26-29) — unless a
 * crypto transform has to see them in the stream's own buffer first (:256). */
int szl_deflater_deflate_view(szl_deflater *d, const uint8_t **p, int64_t *n);
/* The caller declares that it takes everything Deflate() offers before it changes a parameter — every Deflate() loop runs until
 * IsNeedingInput, as DeflaterOutputStream.Deflate() does (CS/DeflaterOutputStream.cs:242-272).  Only then is SetLevel / SetStrategy with
 * 16 KiB or more of input pending answered: where the reference's engine stands at such a call depends on how much output was taken
 * (C/DeflaterEngine.cs:126-139), and this object — which compresses at Flush() / Finish() — cannot observe that.  Without the declaration
 * such a call returns SZL_E_UNSUPPORTED (NotSupportedException) instead of bytes that may differ from the reference's; Flush() first, or
 * declare.  The device-aware stream classes declare it for the Deflater they drive.  Survives Reset(). */
int szl_deflater_caller_drains(szl_deflater *d, int on);
/* (test tap) parts the object's last segment was parsed in while the caller was still writing; 0: compressed in one piece at Flush() / Finish() */
int szl_deflater_debug_pipe_parts(const szl_deflater *d);
int szl_deflater_needs_input(const szl_deflater *d);                       /* IsNeedingInput   C/Deflater.cs:285 */
int szl_deflater_is_finished(const szl_deflater *d);                       /* IsFinished       C/Deflater.cs:271 */
int64_t szl_deflater_total_in(const szl_deflater *d);                      /* TotalIn          C/Deflater.cs:226 */
int64_t szl_deflater_total_out(const szl_deflater *d);                     /* TotalOut         C/Deflater.cs:237 */
uint32_t szl_deflater_adler(const szl_deflater *d);                        /* Adler            C/Deflater.cs:215 */
/* CRC-32 of the input given so far, kept on the device beside the compression (what GZipOutputStream / ZipOutputStream accumulate on the
 * CPU over every Write: S/GZip/GzipOutputStream.cs:210, S/Zip/ZipOutputStream.cs:700): a device-aware container stream switches it on
 * before the first SetInput and reads it where the reference reads crc.Value.  Off by default; 0 when off. */
int szl_deflater_enable_crc32(szl_deflater *d, int on);
uint32_t szl_deflater_crc32(const szl_deflater *d);

/* ------------------------------------------------------------------------------------------
 * Batch / device-resident entry points (SURVEY §8b "one-shot batch entry points"): the fast path
 * for config 2 (one huge stream), config 3 (many small streams, feeds
 * ZipOutputStream.PutNextPassthroughEntry S/Zip/ZipOutputStream.cs:313) and bench.py.
 * Each stream is compressed exactly as `new Deflater(level, nowrap)` + SetInput(all) + Finish()
 * would (bit-identical output), optionally with CRC-32 / Adler-32 of the input computed on device.
 * ------------------------------------------------------------------------------------------ */
typedef struct szl_stream {
    uint64_t in_off;    /* byte offset of this stream's input inside the input buffer */
    uint64_t in_len;
    uint64_t out_off;   /* byte offset of this stream's output region inside the output buffer */
    uint64_t out_cap;   /* size of that region; must be >= szl_deflate_bound(in_len) */
    uint64_t out_len;   /* [out] compressed bytes written */
    uint32_t crc32;     /* [out] Crc32.Value of the input if SZL_F_CRC32 */
    uint32_t adler32;   /* [out] Adler32.Value of the input if SZL_F_ADLER32 or zlib framing */
    int32_t status;     /* [out] per-stream szl_status */
    uint32_t reserved;  /* [in] SZL_F_GZIP: MTIME field of the member header */
    uint64_t in_consumed; /* [out] inflate: compressed bytes consumed (== Inflater.TotalIn at IsFinished) */
} szl_stream;

enum { SZL_F_NOWRAP = 1, SZL_F_CRC32 = 2, SZL_F_ADLER32 = 4, SZL_F_SYNC_FLUSH_BEFORE_FINISH = 8,
       SZL_F_GZIP = 16 /* deflate only: emit a complete RFC 1952 member per stream exactly as GZipOutputStream would with
                          ModifiedTime = streams[i].reserved seconds since the epoch (S/GZip/GzipOutputStream.cs:315-375) */ };

uint64_t szl_deflate_bound(uint64_t in_len);   /* worst-case compressed size the device path may write */

typedef struct szl_engine szl_engine;           /* owns device workspace; reusable across calls; one thread at a time */
szl_engine *szl_engine_create(void);
void szl_engine_destroy(szl_engine *e);

/* Device-resident: d_in/d_out are device pointers; nothing crosses PCIe except the stream table.
 * `hip_stream` is a hipStream_t (NULL = default stream).  Synchronous on return.
 * Only the streams' own regions d_out[out_off, out_off + out_cap) are written (zero-filled, then encoded into); bytes of
 * d_out before, between and after them are left untouched, so a caller may pre-place container framing (zip local
 * headers) around the regions. */
int szl_deflate_batch_device(szl_engine *e, const void *d_in, void *d_out, szl_stream *streams, size_t n_streams,
                             int level, int strategy, unsigned flags, void *hip_stream);
/* Host buffers: copies H2D, runs the same pipeline, copies D2H. */
int szl_deflate_batch_host(szl_engine *e, const void *h_in, void *h_out, szl_stream *streams, size_t n_streams,
                           int level, int strategy, unsigned flags);

/* Several devices of one node behind the same call (SURVEY §8e "shard per GPU": ZipOutputStream entries S/Zip/ZipOutputStream.cs:494,
 * gzip members S/GZip/GzipInputStream.cs:353-357): the streams are cut into n_dev contiguous groups of about equal input bytes and
 * group g is compressed on devices[g] by its own host thread and engine; only a group's own bytes travel to its device.  Results
 * are identical to the single-device calls (every stream is independent).  The same ordinal may appear more than once.
 * ONE stream (n_streams == 1, levels 5-9, at least SZL_PART_MIN_KIB = 64 MiB per device) is not left to a single device: it is
 * cut into position ranges (SZL_PART_UNITS = 4 per device) that the devices take one after the other as they become free — a
 * device with cheaper bytes takes more of them — each device runs the match search and the parse of its ranges (one Deflater
 * lifetime, S/GZip/GzipOutputStream.cs:87, has no other parallel form), the host checks that each range's parse is entered
 * where the previous one leaves it, and devices[0] collects the tokens while later ranges still run (peer copies where the
 * devices reach each other, through the host otherwise) and builds the blocks.  Same bytes as one device.
 * These two entry points take no engine handle: their engines live in a process-wide pool and the calls SERIALISE on it
 * (a second caller waits for the first). */
int szl_deflate_batch_multi_host(const int *devices, int n_dev, const void *h_in, void *h_out, szl_stream *streams, size_t n_streams,
                                 int level, int strategy, unsigned flags);
/* ONE stream over several devices with the input ALREADY RESIDENT: d_in[g] = the input arena in the memory of devices[g] (the caller
 * uploaded the stream to every device once), d_out0 = the output arena on devices[0] (stream->out_off / out_cap inside it).  The same
 * position-range units, hand-over check and token gather as szl_deflate_batch_multi_host's one-stream path — without its host
 * buffers, i.e. without PCIe in the call (bench.py --mode strong times this).  Levels 5-9; same bytes as one engine. */
int szl_deflate_stream_multi_device(const int *devices, int n_dev, const void *const *d_in, void *d_out0, szl_stream *stream,
                                    int level, int strategy, unsigned flags);
/* The multi-device entry points keep one engine per device slot (work space included) between calls, and the streaming objects
 * (szl_deflater / szl_inflater) give their engine — work space and long device buffers included — to a process-wide pool when they are
 * destroyed, from which the next object takes it (SZL_ENGINE_POOL idle engines, default 2, 0 = none: a GZipOutputStream makes a new
 * Deflater per stream, S/GZip/GzipOutputStream.cs:87, and allocating ~19 bytes of device memory per input byte cost its first Finish()
 * 20-800 ms).  This frees both. */
int szl_multi_release(void);
/* Memory the library holds for objects that do not exist any more (round 6).  When the last szl_deflater / szl_inflater is destroyed the
 * idle engines' work space and the pool of pinned blocks shrink to SZL_IDLE_KEEP_MIB each (1024) by themselves once no object has existed
 * for SZL_IDLE_TRIM_MS (2000; a caller who makes one Deflater per stream keeps its buffers); szl_trim() frees ALL of
 * it — idle engines, the multi-device slots, the pinned pool — e.g. from a host's own idle handler.  Live objects are not touched. */
int szl_trim(void);
int szl_inflate_batch_multi_host(const int *devices, int n_dev, const void *h_in, void *h_out, szl_stream *streams, size_t n_streams,
                                 unsigned flags);

/* Per-stage timing of the last batch call on this engine, milliseconds measured with HIP events on the
 * engine's stream (SURVEY §5 "per-stage hipEvent timing exported through the C ABI"). */
typedef struct szl_timing {
    float total_ms, checksum_ms, links_ms, match_ms, parse_ms, blocks_ms, encode_ms;
    uint64_t in_bytes, out_bytes, tokens, blocks, ranges_unmerged, fallback_walks;
    float inflate_ms;   /* last szl_inflate_batch_* call: k_inflate time (HIP events) */
    float pilot_ms;     /* stage-B pilot (sample of tiles + read-back) when SZL_MATCH_MODE = 2 asks for one (the default is the full search
                           without a pilot since round 3); in total_ms, not in match_ms */
    uint32_t links_guard_trips; /* times (this process) stage A's ticket form was caught storing a link out of order — by the check every
                                   exchange carries or by the sampled first-principles guard — and the call was run again with the form that
                                   needs no ordering assumption.  0 on every device seen so far; not silent if it ever is not */
} szl_timing;
int szl_engine_last_timing(const szl_engine *e, szl_timing *t);

/* Debug/parity taps (used by tests to diff intermediates against oracle/szl_model.c): copy the
 * last call's intermediates of stream 0 to host arrays (any pointer may be NULL). */
int szl_engine_debug_fetch(szl_engine *e, uint16_t *link, uint32_t *m2, uint32_t *mq, size_t n_positions,
                           uint32_t *tokens, size_t tok_cap, size_t *n_tokens);

/* Parity tap: choose the form of stage B (FindLongestMatch over all positions = 0, only over the positions a parse can
 * reach = 1, pilot decides = 2, library default = -1; any other value only queries).  Returns 1 if the last call used the
 * on-demand form.  Results are identical either way (C/DeflaterEngine.cs:474-612 is restated by both). */
int szl_engine_debug_match_mode(szl_engine *e, int mode);

/* Experiment / parity knob: sets a named tuning value for this process (the same names are read from the environment as a
 * fallback): SZL_MATCH_KERNEL, SZL_LINKS, SZL_FTH2, SZL_VTH2, SZL_QKEEP, SZL_VKEEP, SZL_DEBUG, ...  Results never depend on them.  value INT_MIN forgets the name again. */
int szl_debug_set(const char *name, int value);

/* Parity tap: device bytes held by the engine's per-position side arrays at the peak of the last deflate call.  A single stream
 * longer than SZL_WINDOW_KIB (default 256 MiB) is processed window by window, so this stays bounded by the window. */
uint64_t szl_engine_debug_workspace(const szl_engine *e);

/* Parity tap: number of chunk jobs the last szl_inflate_batch_* call decoded single members with (0: every stream went through
 * the one-wavefront-per-stream decoder).  In calls of up to SZL_INF_PAR_MAX_STREAMS (default 1024) streams, every member of
 * SZL_INF_PAR_MIN_KIB (default 512) compressed KiB or more is decoded by one wavefront per chunk of 1/32 of its compressed
 * bytes (16 KiB .. SZL_INF_CHUNK_KIB, default 128). */
uint32_t szl_engine_debug_par_jobs(const szl_engine *e);
/* test tap: the form of stage B's instruction text the engine's last launch of the full search ran (0 / 1, 2 = every tile chose): the engine
 * picks it per launch from a sample of the call's prev[] hops (csrc/szl_engine.hip Engine::pick_text_form; no counterpart in the reference,
 * whose FindLongestMatch, C/DeflaterEngine.cs:474-612, both forms restate bit for bit). */
int szl_engine_debug_text_form(const szl_engine *e);

/* Parity tap: the code lengths DeflaterHuffman.Tree.BuildTree + BuildLength (C/DeflaterHuffman.cs:196-329, :475-579) give `n` frequency
 * vectors of `num_symbols` entries each (min_codes / max_length as the three trees have them: 257 / 15, 1 / 15, 4 / 7), as stage D's
 * wavefront-per-tree build computes them — compared with the oracle on histograms no token stream would produce. */
int szl_debug_tree_lengths(const int32_t *freqs, int n, int num_symbols, int min_codes, int max_length, uint8_t *lengths_out, int32_t *num_codes_out);

/* Parity tap (host arithmetic only): stored-block list of a level-0 stream fed as `chunks`; rows: abs_off, len, last. */
int szl_debug_stored_layout(const uint64_t *chunks, size_t nchunks, int flush_before_finish, uint64_t *rows, size_t cap_rows, size_t *n_rows);
/* Test tap (host only, no device): the copy SetInput makes into pinned memory — on several cores when the piece is long
 * (SZL_COPY_THREADS) — applied to plain memory. */
int szl_debug_host_copy(void *dst, const void *src, size_t n);

/* Parity tap: block table of the last call; rows of 8 x uint64:
 * type, last, ntok, bit_start, opt_len, static_len, in_len, hdr_bits. */
int szl_engine_debug_blocks(szl_engine *e, uint64_t *rows, size_t cap_rows, size_t *n_rows);

/* ------------------------------------------------------------------------------------------
 * Inflater — replaces ICSharpCode.SharpZipLib.Zip.Compression.Inflater (C/Inflater.cs)
 * ------------------------------------------------------------------------------------------ */
typedef struct szl_inflater szl_inflater;
szl_inflater *szl_inflater_create(int no_header);                           /* Inflater(bool)   C/Inflater.cs:156-180 */
void szl_inflater_destroy(szl_inflater *s);
int szl_inflater_reset(szl_inflater *s);                                    /* Reset()          C/Inflater.cs:188 */
int szl_inflater_set_input(szl_inflater *s, const uint8_t *p, int n);       /* SetInput         C/Inflater.cs:629 */
int szl_inflater_set_dictionary(szl_inflater *s, const uint8_t *p, int n);  /* SetDictionary    C/Inflater.cs:563 */
int szl_inflater_inflate(szl_inflater *s, uint8_t *out, int count);         /* Inflate          C/Inflater.cs:715 */
int szl_inflater_needs_input(const szl_inflater *s);                        /* IsNeedingInput   C/Inflater.cs:783 */
int szl_inflater_needs_dictionary(const szl_inflater *s);                   /* IsNeedingDictionary :794 */
int szl_inflater_is_finished(const szl_inflater *s);                        /* IsFinished       C/Inflater.cs:806 */
int szl_inflater_remaining_input(const szl_inflater *s);                    /* RemainingInput   C/Inflater.cs:878 */
int64_t szl_inflater_total_in(const szl_inflater *s);                       /* TotalIn          C/Inflater.cs:862 */
int64_t szl_inflater_total_out(const szl_inflater *s);                      /* TotalOut         C/Inflater.cs:848 */
uint32_t szl_inflater_adler(const szl_inflater *s);                         /* Adler            C/Inflater.cs:823 */
/* --- the device-aware InflaterInputBuffer / InflaterInputStream (CS/InflaterInputStream.cs:14-330, :342-700; INTEGRATION.md file 3) ---
 * The reference's buffer class reads `bufferSize` bytes (default 4096, :22, :342-358; GZipInputStream passes 4096,
 * S/GZip/GzipInputStream.cs:72) from the base stream and hands them to Inflater.SetInput.  One wavefront decodes such a piece at
 * ~17 MiB/s; the chunk-parallel decoder needs megabytes per SetInput.  The device-aware buffer class therefore reads AHEAD (16 MiB
 * unless the constructor asked for more) into a buffer of pinned host memory — obtained here — and everything else of the class
 * (Available, RawData/RawLength, ClearText, ReadLeByte .. ReadLeLong, ReadRawBuffer, ReadClearTextBuffer, CryptoTransform) works on
 * that buffer as it did on the small one, so GZipInputStream / ZipInputStream still find their trailers in it.
 *   szl_host_alloc / szl_host_free: a buffer of pinned host memory (a shim whose runtime owns the array pins it and calls
 *   szl_host_register / szl_host_unregister instead).  A SetInput from such a buffer is taken WITHOUT a host copy when the object holds
 *   no older input: the object keeps the pointer, as the reference's StreamManipulator keeps the caller's array
 *   (CS/StreamManipulator.cs:244-262), the device reads it by DMA, and the few bytes the decoder leaves unconsumed are copied away
 *   before IsNeedingInput turns true (the moment the reference allows the caller to refill the array). */
void *szl_host_alloc(size_t n);
void szl_host_free(void *p);
int szl_host_register(void *p, size_t n);
int szl_host_unregister(void *p);
/* The object stops referring to the caller's pinned buffer now (what it has not consumed moves into its own memory): called by the
 * stream shim before it frees or refills a buffer out of turn — Dispose() of a stream whose Inflater lives on (InflaterPool). */
int szl_inflater_detach_input(szl_inflater *s);
/* Hint of the stream shim, given with every SetInput: more != 0 — the buffer just given was filled completely, so more input follows
 * (InflaterInputBuffer.Fill read its whole length, CS/InflaterInputStream.cs:115-128).  A piece decoded by the chunk-parallel decoder then
 * ends on its last block boundary and the object asks for input at once instead of running one wavefront over the cut block behind it;
 * the next piece starts on that boundary.  more == 0 takes the promise back (the base stream has ended): returns 1 if a remainder was
 * waiting — the next Inflate() decodes it, so a truncated stream still delivers every byte it holds — else 0.  Never set: the object
 * decodes everything it is given before it asks for more, as the reference does. */
int szl_inflater_expect_more(szl_inflater *s, int more);
/* CRC-32 of the bytes handed out by Inflate() so far, kept on the device beside the decode (what GZipInputStream / ZipInputStream
 * accumulate on the CPU over every buffer they return: S/GZip/GzipInputStream.cs:141, S/Zip/ZipInputStream.cs:673): a device-aware
 * container stream switches it on before the first SetInput (and again after Reset: it stays on) and reads it where the reference
 * reads crc.Value.  Off by default; 0 when off. */
int szl_inflater_enable_crc32(szl_inflater *s, int on);
uint32_t szl_inflater_crc32(const szl_inflater *s);
/* Parity / measurement tap: pieces of this streaming Inflater's input that went to the chunk-parallel decoder (a SetInput of 2 MiB or
 * more — InflaterInputStream with a large buffer, CS/InflaterInputStream.cs:342-396 — is not decoded by one wavefront) */
uint32_t szl_inflater_debug_bulk_calls(const szl_inflater *s);
/* Measurement tap: wall-clock milliseconds this object has spent, by part — [0] SetInput, [1] upload of long pieces, [2] their decode
 * (finder, symbol pass, windows, bytes), [3] their download, [4] checksums, [5] all decoder steps together ([1]..[4] are inside),
 * [6] the copies out of Inflate(), [7] the one-wavefront steps alone. */
int szl_inflater_debug_times(const szl_inflater *s, double *ms8);

/* Batch inflate of independent raw-deflate / zlib streams (zip entries, gzip members): one
 * wavefront per stream.  streams[i].in_* = compressed bytes, out_* = region for the decompressed
 * bytes, out_len = [out] decompressed size, in_consumed = [out] compressed bytes consumed
 * (== Inflater.TotalIn at IsFinished).  flags: SZL_F_NOWRAP, SZL_F_CRC32 (crc of OUTPUT), SZL_F_ADLER32. */
int szl_inflate_batch_device(szl_engine *e, const void *d_in, void *d_out, szl_stream *streams, size_t n_streams,
                             unsigned flags, void *hip_stream);
int szl_inflate_batch_host(szl_engine *e, const void *h_in, void *h_out, szl_stream *streams, size_t n_streams,
                           unsigned flags);

#ifdef __cplusplus
}
#endif
#endif /* SZL_H */
