// szl_kernels_match2.hip — stage B (FindLongestMatch for every position), the form of the full search used by default.
//
// Reference being restated: FindLongestMatch, C/DeflaterEngine.cs:474-612 (same results as k_match in
// szl_kernels_match.hip: M2 = walk entered with matchLen 2 and the full max_chain budget, Mq = that walk's state after
// max_chain>>2 candidates).
//
// Why a second form (measurements: profiles/r02/pmc_stage_b.json, lab_*.log).  k_match is bound by VALU issue: 43.8 VALU +
// 25.7 SALU wave-instructions per position with the SIMDs' VALU slots ~100 % busy — and only 42 % of the lanes take part in
// an average chain step, because a lane can only work on the phase its one position is in.  Two things do NOT fix that
// (both built and measured this round): exchanging a lane's two positions so that either can serve the running phase
// (the selects of the exchange cost more than the lanes gained), and cheaper steps alone (with 4 waves per SIMD — all the
// 147 KiB window leaves room for — a wave that needs one LDS round trip per step stalls instead).  What does:
//   * the chain walk (QUICK) and the byte compare (VERIFY) are hand-written loops in which a lane that leaves the phase is
//     cleared from the exec mask by v_cmpx, so the surviving lanes update their state in place: 7 VALU per chain step
//     instead of 13 (no per-step select of every state variable, no mode bookkeeping; the states are SGPR lane masks);
//   * every lane holds TWO walks whose LDS reads are in flight together, software-pipelined across steps, so a wave waits
//     for one LDS round trip per two steps;
//   * VERIFY compares 8 bytes per step from ALIGNED dwords + v_alignbyte (32-bit LDS reads at byte addresses are replayed:
//     SQ_LDS_UNALIGNED_STALL made the LDS pipe the bottleneck in the first version);
//   * a freshly fetched position goes straight to VERIFY with its first candidate (same 3-byte hash: the scan_end test at
//     offset 2 all but always passes), saving one chain step per position.
//   * QUICK tests the two bytes the reference tests first (scan_end and scan_end1, :505-506) instead of one: 40 % fewer
//     candidates reach VERIFY on text, 3x fewer on logs at level 9 (no gain on text — a third LDS read per step pays for it —
//     but 271 -> 238 ms per GiB of logs).
// Result: 89.9 -> 64 ms per GiB of text at level 6 (20.1 VALU + 15.7 SALU + 6.0 LDS per position), 55 ms after the tile's positions
// went out in slices of 128 instead of 512 and the thresholds were swept again (VERIFY as soon as two contexts wait);
// 475 -> 201 ms per GiB of logs at level 9 (full search), bit-identical tables.
//
// Built, measured and dropped again this round (sources in the history, logs under profiles/r02): four contexts per lane
// (lab_s11: 99 ms per GiB — every issue slot is paid per context whatever its lane count); a run-ahead engine in which a
// candidate that passes the filter waits in a per-context pending slot while the walk goes on (lab_s13/s14: QUICK runs with
// 54 % of the contexts instead of 40 %, but the per-pass cost of VERIFY/COMPLETE grows by the same amount: 65-70 ms).  With
// VALU ~80 %, SALU ~70 % and the LDS pipe ~60 % busy at four waves per SIMD every variant lands at 64-66 ms: the next step
// has to remove chain steps (34 per position), not re-balance them.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdlib>
#include <cstdint>
#include "szl_internal.h"

namespace szl {
int knob(const char *name, int dflt);

__device__ __forceinline__ int64_t base_of2(int64_t s_abs) { // window base of an iteration starting at s (App. A.2; C/DeflaterEngine.cs:371,:771,:93)
    int64_t idx = s_abs + 1;
    if (idx <= 65273) return 0;
    return ((idx - 65273 + 32767) >> 15) << 15;
}
__device__ __forceinline__ uint32_t load_u32_unaligned2(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// A tile costs ~67 us beyond its walks (staging of the history, and the wait for the longest walks at its end: profiles/r02/
// lab_s46_tile_length.log), so the tile is as long as the LDS of a CU allows: 3 bytes per position of history + tile.
enum : int { B2_THREADS = 1024, B2_TILE = 21504 };
enum : int { SZL_B9_DEFAULT = 1 };   // k_match9 (szl_kernels_match9.hip); 0 = k_match4 below
enum : int { B2_DATA_BYTES = B_HIST + B2_TILE + B_TAIL + 8, B2_LINKS = B_HIST + B2_TILE };
enum : int { B2_LDS_BYTES = B2_DATA_BYTES + B2_LINKS * 2 + 32 };   // (+ the tile counter and, in the debug build, two time stamps)
static_assert(B2_LDS_BYTES <= 160 * 1024 && B2_DATA_BYTES % 16 == 0 && B2_TILE % 64 == 0, "the window must fit the CU's LDS");

#if SZL_LAB   // (laboratory library only from here to launch_match2: k_match4 — stage B's engine of rounds 2-3, superseded by k_match9 — and the ring k_match8.
              // Round 5 took k_match4 out of the product library: the only form of the full search that ships is k_match9, szl_kernels_match9.hip)
typedef __attribute__((address_space(3))) uint8_t lds_u8;

// ---- the hand-written engine ---------------------------------------------------------------------------------------------
// Every lane holds TWO walks (contexts A and B); a context is in one of four states kept as wave-level lane masks in SGPRs:
// q (QUICK: walking its chain), v (VERIFY: comparing a candidate), d (DONE: results wait to be stored), none (NEED: free).
// The engine runs QUICK and VERIFY phases until enough contexts are free for a FETCH (done in C++), with both contexts'
// LDS reads of a step in flight together: per step-pair one LDS round trip instead of two (the kernel is otherwise bound by
// that latency: 16 waves per CU is all the 147 KiB window leaves room for).
//   QUICK step of a context (C/DeflaterEngine.cs:502-507,:609): read the candidate's scan_end byte and its prev[] hop; v_cmpx
//   clears from exec the lanes whose byte matches (-> VERIFY), whose next candidate is out of the window and whose chain budget
//   is spent (-> DONE); the lanes left move to the next candidate in place.
//   VERIFY step (:515-591): 8 bytes of candidate and position per step from the byte address; exec keeps the lanes that matched
//   all 8 and are below `cap`.  Completed lanes run :593-609.
#define SZL_Q_ISSUE(X) \
    "v_lshl_add_u32 %[t0" #X "], %[cl" #X "], 1, %[lbase]\n\t" \
    "v_add3_u32 %[t1" #X "], %[cl" #X "], %[best" #X "], %[dbm1]\n\t" \
    "ds_read_u16 %[t0" #X "], %[t0" #X "]\n\t" \
    "ds_read_u8 %[t2" #X "], %[t1" #X "]\n\t" \
    "ds_read_u8 %[t1" #X "], %[t1" #X "] offset:1\n\t"
// t0 = prev[] hop, t2 / t1 = the candidate's bytes at best_len - 1 / best_len (scan_end1, scan_end: :505-506).  Testing both
// sends 40 % fewer candidates to VERIFY on text than scan_end alone (3.55 -> 2.15 per position; 18.3 -> 5.7 on logs at level 9).
// Two byte reads on purpose: ONE ds_read_u16 at the (odd or even) byte address was measured at 103 instead of 64 ms per GiB
// (profiles/r02/lab_s21_unaligned_u16.log) — every LDS access that is not naturally aligned is replayed, not only 32-bit ones.
// The lanes that stay advance in place; a lane that leaves because the bytes match keeps `cl` (the candidate to compare).
#define SZL_Q_FINISH(X) \
    "v_lshl_or_b32 %[t1" #X "], %[t1" #X "], 8, %[t2" #X "]\n\t" \
    "v_cmpx_ne_u32 vcc, %[pb" #X "], %[t1" #X "]\n\t" \
    "v_sub_u32 %[cl" #X "], %[cl" #X "], %[t0" #X "]\n\t" \
    "v_cmpx_ge_i32 vcc, %[cl" #X "], %[mincl" #X "]\n\t" \
    "v_subrev_co_u32 %[left" #X "], vcc, 1, %[left" #X "]\n\t" \
    "s_andn2_b64 exec, exec, vcc\n\t" \
    "s_mov_b64 %[m" #X "], exec\n\t"
// the same for the last step of an iteration: exec is not used again before it is reloaded
#define SZL_Q_FINISH_LAST(X) \
    "v_lshl_or_b32 %[t1" #X "], %[t1" #X "], 8, %[t2" #X "]\n\t" \
    "v_cmpx_ne_u32 vcc, %[pb" #X "], %[t1" #X "]\n\t" \
    "v_sub_u32 %[cl" #X "], %[cl" #X "], %[t0" #X "]\n\t" \
    "v_cmpx_ge_i32 vcc, %[cl" #X "], %[mincl" #X "]\n\t" \
    "v_subrev_co_u32 %[left" #X "], vcc, 1, %[left" #X "]\n\t" \
    "s_andn2_b64 %[m" #X "], exec, vcc\n\t"
// leavers of context X: lanes of q that are no longer in m; the byte read last says which way (a match wins over the chain end)
#define SZL_Q_CLASSIFY(X) \
    "s_andn2_b64 exec, %[q" #X "], %[m" #X "]\n\t" \
    "v_cmp_eq_u32 vcc, %[pb" #X "], %[t1" #X "]\n\t" \
    "v_mov_b32 %[off" #X "], 0\n\t" \
    "s_or_b64 %[v" #X "], %[v" #X "], vcc\n\t" \
    "s_andn2_b64 %[sc], exec, vcc\n\t" \
    "s_or_b64 %[d" #X "], %[d" #X "], %[sc]\n\t" \
    "s_mov_b64 %[q" #X "], %[m" #X "]\n\t"
// 8 bytes of each side from three ALIGNED dwords + v_alignbyte (a 32-bit LDS read at a byte address that is not a multiple of
// 4 is replayed: SQ_LDS_UNALIGNED_STALL made the LDS pipe the bottleneck when this loop read at the byte address directly)
#define SZL_V_ISSUE(X) \
    "v_add3_u32 %[t0" #X "], %[cl" #X "], %[off" #X "], %[dbase]\n\t" \
    "v_add3_u32 %[t1" #X "], %[p" #X "], %[off" #X "], %[pbase]\n\t" \
    "v_and_b32 %[t4" #X "], -4, %[t0" #X "]\n\t" \
    "v_and_b32 %[t5" #X "], -4, %[t1" #X "]\n\t" \
    "ds_read_b32 %[t2" #X "], %[t4" #X "]\n\t" \
    "ds_read_b32 %[t3" #X "], %[t4" #X "] offset:4\n\t" \
    "ds_read_b32 %[t4" #X "], %[t4" #X "] offset:8\n\t" \
    "ds_read_b32 %[t6" #X "], %[t5" #X "]\n\t" \
    "ds_read_b32 %[t7" #X "], %[t5" #X "] offset:4\n\t" \
    "ds_read_b32 %[t5" #X "], %[t5" #X "] offset:8\n\t"
// candidate: dwords t2,t3,t4 shifted by t0[1:0] bytes; position: t6,t7,t5 shifted by t1[1:0]
#define SZL_V_FINISH(X) \
    "v_alignbyte_b32 %[t2" #X "], %[t3" #X "], %[t2" #X "], %[t0" #X "]\n\t" \
    "v_alignbyte_b32 %[t3" #X "], %[t4" #X "], %[t3" #X "], %[t0" #X "]\n\t" \
    "v_alignbyte_b32 %[t6" #X "], %[t7" #X "], %[t6" #X "], %[t1" #X "]\n\t" \
    "v_alignbyte_b32 %[t7" #X "], %[t5" #X "], %[t7" #X "], %[t1" #X "]\n\t" \
    "v_xor_b32 %[t2" #X "], %[t2" #X "], %[t6" #X "]\n\t" \
    "v_xor_b32 %[t0" #X "], %[t3" #X "], %[t7" #X "]\n\t" \
    "v_ffbl_b32 %[t2" #X "], %[t2" #X "]\n\t" \
    "v_ffbl_b32 %[t0" #X "], %[t0" #X "]\n\t" \
    "v_or_b32 %[t0" #X "], 32, %[t0" #X "]\n\t" \
    "v_min_u32 %[t2" #X "], %[t2" #X "], %[t0" #X "]\n\t" \
    "v_lshrrev_b32 %[t2" #X "], 3, %[t2" #X "]\n\t" \
    "v_min_u32 %[t2" #X "], 8, %[t2" #X "]\n\t" \
    "v_add_u32 %[off" #X "], %[off" #X "], %[t2" #X "]\n\t" \
    "v_cmpx_eq_u32 vcc, 8, %[t2" #X "]\n\t" \
    "v_cmpx_lt_i32 vcc, %[off" #X "], %[cap" #X "]\n\t" \
    "s_mov_b64 %[m" #X "], exec\n\t"
// :593-609 for the lanes of v that are no longer in m (comparison complete)
#define SZL_V_COMPLETE(X) \
    "s_andn2_b64 %[cm], %[v" #X "], %[m" #X "]\n\t" \
    "s_mov_b64 exec, %[cm]\n\t" \
    "v_lshl_add_u32 %[t0" #X "], %[cl" #X "], 1, %[lbase]\n\t" \
    "ds_read_u16 %[t0" #X "], %[t0" #X "]\n\t"                              /* prev[] hop of this candidate */ \
    "v_min_i32 %[t2" #X "], %[off" #X "], %[cap" #X "]\n\t"                /* L */ \
    "v_cmp_gt_i32 %[sc], %[t2" #X "], %[best" #X "]\n\t"                   /* strictly longer: the new best */ \
    "s_mov_b64 exec, %[sc]\n\t" \
    "v_mov_b32 %[best" #X "], %[t2" #X "]\n\t" \
    "v_sub_u32 %[t1" #X "], %[p" #X "], %[cl" #X "]\n\t" \
    "v_add_u32 %[t1" #X "], %[bhist], %[t1" #X "]\n\t"                       /* distance = (p + B_HIST) - cl */ \
    "v_lshl_or_b32 %[res2" #X "], %[t1" #X "], 16, %[t2" #X "]\n\t" \
    "v_cmp_ge_i32 vcc, %[left" #X "], %[snap]\n\t"                           /* seen by the quarter-budget walk too (:495) */ \
    "v_cndmask_b32 %[resq" #X "], %[resq" #X "], %[res2" #X "], vcc\n\t" \
    "v_add3_u32 %[t1" #X "], %[p" #X "], %[t2" #X "], %[pbm1]\n\t" \
    "ds_read_u8 %[t3" #X "], %[t1" #X "]\n\t"                               /* scan_end1 / scan_end bytes for the new best_len */ \
    "ds_read_u8 %[t1" #X "], %[t1" #X "] offset:1\n\t" \
    "v_cmp_ge_i32 %[sc], %[t2" #X "], %[nice" #X "]\n\t"                   /* >= niceLength: stop (:603) */ \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_lshl_or_b32 %[pb" #X "], %[t1" #X "], 8, %[t3" #X "]\n\t" \
    "s_mov_b64 exec, %[cm]\n\t" \
    "v_sub_u32 %[t0" #X "], %[cl" #X "], %[t0" #X "]\n\t"                  /* next candidate */ \
    "v_cmp_lt_i32 vcc, %[t0" #X "], %[mincl" #X "]\n\t" \
    "s_or_b64 %[sc], %[sc], vcc\n\t" \
    "v_cmp_eq_u32 vcc, 0, %[left" #X "]\n\t" \
    "s_or_b64 %[sc], %[sc], vcc\n\t"                                         /* sc = lanes whose walk ends here */ \
    "v_mov_b32 %[off" #X "], 0\n\t" \
    "s_or_b64 %[d" #X "], %[d" #X "], %[sc]\n\t" \
    "s_andn2_b64 exec, %[cm], %[sc]\n\t" \
    "v_mov_b32 %[cl" #X "], %[t0" #X "]\n\t" \
    "v_add_u32 %[left" #X "], -1, %[left" #X "]\n\t" \
    "s_or_b64 %[q" #X "], %[q" #X "], exec\n\t" \
    "s_mov_b64 %[v" #X "], %[m" #X "]\n\t"

// The engine's instruction text, shared by k_match4 (plain and with lab counters) and k_match8.  QFIN / QFINL / VCOMP name the
// QUICK-step and COMPLETE macros of the kernel (window indices or ring addresses); KQ / KV / KC are empty or the lab counters.
#define SZL_ENGINE_TEXT(QFIN, QFINL, VCOMP, KQ, KV, KC) \
    "s_mov_b64 %[sv], exec\n" \
    "10:\n\t"                                           /* ---- census */ \
    "s_or_b64 %[sc], %[qA], %[vA]\n\t" \
    "s_bcnt1_i32_b64 %[n0], %[sc]\n\t" \
    "s_or_b64 %[sc], %[qB], %[vB]\n\t" \
    "s_bcnt1_i32_b64 %[n1], %[sc]\n\t" \
    "s_add_u32 %[n0], %[n0], %[n1]\n\t"               /* busy contexts */ \
    "s_cmp_le_u32 %[n0], %[bexit]\n\t" \
    "s_cbranch_scc1 19f\n\t" \
    "s_bcnt1_i32_b64 %[n1], %[vA]\n\t" \
    "s_bcnt1_i32_b64 %[n2], %[vB]\n\t" \
    "s_add_u32 %[n1], %[n1], %[n2]\n\t"               /* contexts waiting for VERIFY */ \
    "s_cmp_ge_u32 %[n1], %[vth]\n\t" \
    "s_cbranch_scc1 14f\n\t" \
    "s_cmp_eq_u32 %[n0], %[n1]\n\t"                    /* nothing in QUICK */ \
    "s_cbranch_scc1 14f\n" \
    /* ---- QUICK phase */ \
    "s_mov_b64 %[mA], %[qA]\n\t" \
    "s_mov_b64 %[mB], %[qB]\n" \
    "11:\n\t" \
    KQ \
    "s_mov_b64 exec, %[mA]\n\t" \
    SZL_Q_ISSUE(A) \
    "s_mov_b64 exec, %[mB]\n\t" \
    SZL_Q_ISSUE(B) \
    "s_mov_b64 exec, %[mA]\n\t" \
    "s_waitcnt lgkmcnt(3)\n\t" \
    QFIN(A) \
    SZL_Q_ISSUE(A)                               /* A's next step is in flight while B finishes */ \
    "s_mov_b64 exec, %[mB]\n\t" \
    "s_waitcnt lgkmcnt(3)\n\t" \
    QFIN(B) \
    SZL_Q_ISSUE(B) \
    "s_mov_b64 exec, %[mA]\n\t" \
    "s_waitcnt lgkmcnt(3)\n\t" \
    QFINL(A) \
    "s_mov_b64 exec, %[mB]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    QFINL(B) \
    "s_bcnt1_i32_b64 %[n0], %[mA]\n\t" \
    "s_bcnt1_i32_b64 %[n1], %[mB]\n\t" \
    "s_add_u32 %[n0], %[n0], %[n1]\n\t" \
    "s_cmp_ge_u32 %[n0], %[qkeep]\n\t" \
    "s_cbranch_scc1 11b\n\t" \
    SZL_Q_CLASSIFY(A) \
    SZL_Q_CLASSIFY(B) \
    "s_branch 10b\n" \
    /* ---- VERIFY phase (with VERIFY running as soon as two contexts wait, one side is often empty: then only the other side's */ \
    /* instructions are issued) */ \
    "14:\n\t" \
    "s_mov_b64 %[mA], %[vA]\n\t" \
    "s_mov_b64 %[mB], %[vB]\n\t" \
    "s_cmp_eq_u64 %[vB], 0\n\t" \
    "s_cbranch_scc1 16f\n\t" \
    "s_cmp_eq_u64 %[vA], 0\n\t" \
    "s_cbranch_scc1 17f\n" \
    "15:\n\t" \
    KV \
    "s_mov_b64 exec, %[mA]\n\t" \
    SZL_V_ISSUE(A) \
    "s_mov_b64 exec, %[mB]\n\t" \
    SZL_V_ISSUE(B) \
    "s_mov_b64 exec, %[mA]\n\t" \
    "s_waitcnt lgkmcnt(6)\n\t" \
    SZL_V_FINISH(A) \
    "s_mov_b64 exec, %[mB]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    SZL_V_FINISH(B) \
    "s_bcnt1_i32_b64 %[n0], %[mA]\n\t" \
    "s_bcnt1_i32_b64 %[n1], %[mB]\n\t" \
    "s_add_u32 %[n0], %[n0], %[n1]\n\t" \
    "s_cmp_ge_u32 %[n0], %[vkeep]\n\t" \
    "s_cbranch_scc1 15b\n\t" \
    KC \
    VCOMP(A) \
    VCOMP(B) \
    "s_branch 10b\n" \
    "16:\n\t"                                          /* only context A has candidates to compare */ \
    "s_mov_b64 exec, %[mA]\n\t" \
    SZL_V_ISSUE(A) \
    "s_waitcnt lgkmcnt(0)\n\t" \
    SZL_V_FINISH(A) \
    "s_bcnt1_i32_b64 %[n0], %[mA]\n\t" \
    "s_cmp_ge_u32 %[n0], %[vkeep]\n\t" \
    "s_cbranch_scc1 16b\n\t" \
    VCOMP(A) \
    "s_branch 10b\n" \
    "17:\n\t"                                          /* only context B */ \
    "s_mov_b64 exec, %[mB]\n\t" \
    SZL_V_ISSUE(B) \
    "s_waitcnt lgkmcnt(0)\n\t" \
    SZL_V_FINISH(B) \
    "s_bcnt1_i32_b64 %[n0], %[mB]\n\t" \
    "s_cmp_ge_u32 %[n0], %[vkeep]\n\t" \
    "s_cbranch_scc1 17b\n\t" \
    VCOMP(B) \
    "s_branch 10b\n" \
    "19:\n\t" \
    "s_mov_b64 exec, %[sv]\n\t"
// lab counters of the phases: loop iterations and the contexts that took part
#define SZL_ENGINE_KQ \
    "s_add_u32 %[kq], %[kq], 1\n\t" \
    "s_bcnt1_i32_b64 %[n2], %[mA]\n\t" \
    "s_add_u32 %[kql], %[kql], %[n2]\n\t" \
    "s_bcnt1_i32_b64 %[n2], %[mB]\n\t" \
    "s_add_u32 %[kql], %[kql], %[n2]\n\t"
#define SZL_ENGINE_KV \
    "s_add_u32 %[kv], %[kv], 1\n\t" \
    "s_bcnt1_i32_b64 %[n2], %[mA]\n\t" \
    "s_add_u32 %[kvl], %[kvl], %[n2]\n\t" \
    "s_bcnt1_i32_b64 %[n2], %[mB]\n\t" \
    "s_add_u32 %[kvl], %[kvl], %[n2]\n\t"
#define SZL_ENGINE_KC \
    "s_add_u32 %[kc], %[kc], 1\n\t" \
    "s_andn2_b64 %[cm], %[vA], %[mA]\n\t" \
    "s_bcnt1_i32_b64 %[n2], %[cm]\n\t" \
    "s_add_u32 %[kcl], %[kcl], %[n2]\n\t" \
    "s_andn2_b64 %[cm], %[vB], %[mB]\n\t" \
    "s_bcnt1_i32_b64 %[n2], %[cm]\n\t" \
    "s_add_u32 %[kcl], %[kcl], %[n2]\n\t"

struct WalkCtx {      // one FindLongestMatch walk in flight (all of it in registers)
    int p;            // tile position being searched
    int cl;           // LDS data index of the current candidate (curMatch)
    int best;         // best_len (:483)
    int left;         // candidates that may still FOLLOW the current one (= the reference's chainLength - 1)
    int off;          // VERIFY: bytes of the candidate already known to match
    int mincl;        // candidates below this LDS index end the walk (limit / window index 0, :609)
    int cap, nice;    // min(258, lookahead), min(niceLength, lookahead) (:479,:485)
    uint32_t pb;      // byte of the position at offset best_len (scan_end, :505)
    uint32_t res2, resq;
};

// Stage the window of one tile into LDS: the bytes of history + tile + lookahead tail, and the links of history + tile.
// (NT: threads of the workgroup; the lab kernels can be launched with fewer than B2_THREADS and pass blockDim.x)
__device__ __forceinline__ void b2_stage_window(uint32_t *sdata32, uint16_t *slink, const uint8_t *d, const uint16_t *lk, int64_t dlo,
                                                int64_t seg_end, int64_t t0, int tlen, const int NT = B2_THREADS) {
    for (int i = threadIdx.x; i < B2_DATA_BYTES / 4; i += NT) {
        int64_t pos = dlo + 4 * (int64_t)i;
        uint32_t w = 0;
        if (pos >= 0 && pos + 4 <= seg_end) w = load_u32_unaligned2(d + pos);
        else {
            for (int k = 0; k < 4; k++) {
                int64_t pk = pos + k;
                if (pk >= 0 && pk < seg_end) w |= (uint32_t)d[pk] << (8 * k);
            }
        }
        sdata32[i] = w;
    }
    for (int i = threadIdx.x; i < B2_LINKS / 2; i += NT) {
        int64_t pos = dlo + 2 * (int64_t)i;
        uint32_t w = 0;
        if (pos >= 0 && pos + 2 <= t0 + tlen) {
            uint16_t a = lk[pos], c = lk[pos + 1];
            w = (uint32_t)a | ((uint32_t)c << 16);
        } else {
            if (pos >= 0 && pos < t0 + tlen) w |= lk[pos];
            if (pos + 1 >= 0 && pos + 1 < t0 + tlen) w |= (uint32_t)lk[pos + 1] << 16;
        }
        // "no previous position" (0) is staged as 0xFFFF: cl - 0xFFFF is below every limit (real links are <= 32767)
        if ((w & 0xFFFFu) == 0) w |= 0xFFFFu;
        if ((w >> 16) == 0) w |= 0xFFFF0000u;
        ((uint32_t *)slink)[i] = w;
    }
}

#include "szl_match4_body.inc"

// ---- k_match8: the same engine fed from a RING ----------------------------------------------------------------------------------
// A tile costs k_match4 ~67 us beyond its walks (profiles/r02/lab_s46_tile_length.log: 52.5 / 69.8 / 101.7 ms per GiB with 16 / 8 / 4 Ki
// tiles): it stages 48 Ki positions to search 16 Ki, and it ends with the workgroup waiting for its longest walks with most lanes
// idle.  A first attempt kept the window and moved it along a stripe behind a barrier, walks in flight moving with it (k_match7, in
// the history; lab_s47_*, lab_s48_*): a move cost as much as a new tile — ~20 us however far it moved — because a wave that waits
// at a barrier has its walks paused while its free lanes cannot be refilled.  Here nothing stops: bytes and links live in a ring of R8_W positions (ring address = position mod R8_W), waves take
// slices of positions from one counter as long as the ring holds their lookahead, and whichever wave finds the ring running low
// stages the next R8_C positions over the oldest ones — allowed as soon as no walk can still reach them, which the waves publish as
// their lowest position in flight.  Ring addresses cost the chain step three more VALU instructions (the hop wraps; the limit
// test becomes a test of the accumulated distance), and save the distance computation when a match is recorded.
// Measured (lab_s49_ring_k_match8.log): bit-exact, 55.0 ms per GiB against k_match4's 53.0.  The loop counters say why: 43 % of the
// engine calls run without positions to hand out and staging is refused 19 times per chunk staged — the ring holds 32.5 Ki positions
// of history plus 20 Ki, and the slowest walk in flight (a full chain: ~100 us, 12 Ki positions of progress for the workgroup) pins
// its history, so the lead the ring can build is what a tile had.  The limit is LDS per workgroup, not the tiling.  Lab form
// (SZL_MATCH_KERNEL=4), kept because it removes the 3x staging traffic (9.6 -> 3.3 GB per GiB) should that ever matter.
enum : int { R8_W = 53248, R8_C = 2048, R8_NCH = R8_W / R8_C, R8_PAD = 288, R8_H = 32768 };
enum : int { R8_LINK_OFF = R8_W + R8_PAD, R8_CTL_OFF = R8_LINK_OFF + R8_W * 2, R8_LDS_BYTES = R8_CTL_OFF + 128 };
static_assert(R8_W % R8_C == 0 && R8_LINK_OFF % 16 == 0 && R8_LDS_BYTES <= 160 * 1024 && R8_H >= B_HIST && R8_PAD >= B_TAIL + 16, "ring layout");
enum : int { R8_COUNTER = 0, R8_STAGED = 1, R8_LOCK = 2, R8_SLOT0 = 4 };

#define SZL8_Q_FINISH_(X, TAIL) \
    "v_lshl_or_b32 %[t1" #X "], %[t1" #X "], 8, %[t2" #X "]\n\t" \
    "v_cmpx_ne_u32 vcc, %[pb" #X "], %[t1" #X "]\n\t" \
    "v_add_u32 %[dist" #X "], %[dist" #X "], %[t0" #X "]\n\t"              /* distance of the next candidate ("no link" is 0xFFFF) */ \
    "v_cmpx_le_u32 vcc, %[dist" #X "], %[maxd" #X "]\n\t"                   /* curMatch > limit (:609) */ \
    "v_sub_u32 %[cl" #X "], %[cl" #X "], %[t0" #X "]\n\t" \
    "v_add_u32 %[t2" #X "], %[ringw], %[cl" #X "]\n\t"                       /* below ring address 0: wraps */ \
    "v_min_u32 %[cl" #X "], %[cl" #X "], %[t2" #X "]\n\t" \
    "v_subrev_co_u32 %[left" #X "], vcc, 1, %[left" #X "]\n\t" \
    TAIL
#define SZL8_Q_FINISH(X) SZL8_Q_FINISH_(X, "s_andn2_b64 exec, exec, vcc\n\t" "s_mov_b64 %[m" #X "], exec\n\t")
#define SZL8_Q_FINISH_LAST(X) SZL8_Q_FINISH_(X, "s_andn2_b64 %[m" #X "], exec, vcc\n\t")
#define SZL8_V_COMPLETE(X) \
    "s_andn2_b64 %[cm], %[v" #X "], %[m" #X "]\n\t" \
    "s_mov_b64 exec, %[cm]\n\t" \
    "v_lshl_add_u32 %[t0" #X "], %[cl" #X "], 1, %[lbase]\n\t" \
    "ds_read_u16 %[t0" #X "], %[t0" #X "]\n\t"                              /* prev[] hop of this candidate */ \
    "v_min_i32 %[t2" #X "], %[off" #X "], %[cap" #X "]\n\t"                /* L */ \
    "v_cmp_gt_i32 %[sc], %[t2" #X "], %[best" #X "]\n\t" \
    "s_mov_b64 exec, %[sc]\n\t" \
    "v_mov_b32 %[best" #X "], %[t2" #X "]\n\t" \
    "v_lshl_or_b32 %[res2" #X "], %[dist" #X "], 16, %[t2" #X "]\n\t" \
    "v_cmp_ge_i32 vcc, %[left" #X "], %[snap]\n\t"                           /* seen by the quarter-budget walk too (:495) */ \
    "v_cndmask_b32 %[resq" #X "], %[resq" #X "], %[res2" #X "], vcc\n\t" \
    "v_add3_u32 %[t1" #X "], %[p" #X "], %[t2" #X "], %[pbm1]\n\t" \
    "ds_read_u8 %[t3" #X "], %[t1" #X "]\n\t" \
    "ds_read_u8 %[t1" #X "], %[t1" #X "] offset:1\n\t" \
    "v_cmp_ge_i32 %[sc], %[t2" #X "], %[nice" #X "]\n\t"                   /* >= niceLength: stop (:603) */ \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_lshl_or_b32 %[pb" #X "], %[t1" #X "], 8, %[t3" #X "]\n\t" \
    "s_mov_b64 exec, %[cm]\n\t" \
    "v_add_u32 %[dist" #X "], %[dist" #X "], %[t0" #X "]\n\t" \
    "v_cmp_gt_u32 vcc, %[dist" #X "], %[maxd" #X "]\n\t" \
    "s_or_b64 %[sc], %[sc], vcc\n\t" \
    "v_cmp_eq_u32 vcc, 0, %[left" #X "]\n\t" \
    "s_or_b64 %[sc], %[sc], vcc\n\t"                                         /* sc = lanes whose walk ends here */ \
    "v_mov_b32 %[off" #X "], 0\n\t" \
    "s_or_b64 %[d" #X "], %[d" #X "], %[sc]\n\t" \
    "s_andn2_b64 exec, %[cm], %[sc]\n\t" \
    "v_sub_u32 %[cl" #X "], %[cl" #X "], %[t0" #X "]\n\t" \
    "v_add_u32 %[t1" #X "], %[ringw], %[cl" #X "]\n\t" \
    "v_min_u32 %[cl" #X "], %[cl" #X "], %[t1" #X "]\n\t" \
    "v_add_u32 %[left" #X "], -1, %[left" #X "]\n\t" \
    "s_or_b64 %[q" #X "], %[q" #X "], exec\n\t" \
    "s_mov_b64 %[v" #X "], %[m" #X "]\n\t"

struct WalkCtx8 {
    int p;            // ring address of the position being searched
    int pg;           // the same position counted from the stripe's origin
    int cl;           // ring address of the current candidate
    uint32_t dist;    // its distance from the position
    uint32_t maxd;    // largest distance a chain candidate may have (:609)
    int best, left, off, cap, nice;
    uint32_t pb, res2, resq;
};

// stage positions [k * R8_C, (k + 1) * R8_C) (counted from `origin`) into their place in the ring; threads tid of nthreads
__device__ __forceinline__ void r8_stage_chunk(uint8_t *smem, const uint8_t *d, const uint16_t *lk, int64_t origin, int k, int64_t seg_end, int64_t link_end,
                                               int tid, int nthreads) {
    uint32_t *sdata32 = (uint32_t *)smem;
    uint32_t *slink32 = (uint32_t *)(smem + R8_LINK_OFF);
    const int ra0 = (k % R8_NCH) * R8_C;
    const int64_t pos0 = origin + (int64_t)k * R8_C;
    if (nthreads == 64 && pos0 >= 0 && pos0 + R8_C <= seg_end && pos0 + R8_C <= link_end) {
        // one wave, chunk inside the stream (the steady state): all six 16-byte loads of a lane are in flight together — a chunk
        // has to arrive faster than the workgroup searches one (~17 us), and one wave stages at a time
        static_assert(R8_C == 2048 && R8_W % 16 == 0 && (R8_LINK_OFF % 16) == 0, "two data and four link loads per lane");
        auto ld16 = [](const void *p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; };
        auto fix = [](uint32_t w) { if ((w & 0xFFFFu) == 0) w |= 0xFFFFu; if ((w >> 16) == 0) w |= 0xFFFF0000u; return w; };
        const uint8_t *dp = d + pos0;
        const uint8_t *lp = (const uint8_t *)(lk + pos0);
        const uint4 d0 = ld16(dp + tid * 16), d1 = ld16(dp + (tid + 64) * 16);
        uint4 l0 = ld16(lp + tid * 16), l1 = ld16(lp + (tid + 64) * 16), l2 = ld16(lp + (tid + 128) * 16), l3 = ld16(lp + (tid + 192) * 16);
        uint4 *sd4 = (uint4 *)smem, *sl4 = (uint4 *)(smem + R8_LINK_OFF);
        sd4[(ra0 >> 4) + tid] = d0; sd4[(ra0 >> 4) + tid + 64] = d1;
        if (ra0 == 0 && tid * 16 < R8_PAD) sd4[(R8_W >> 4) + tid] = d0;       // (R8_PAD <= 1024: within the first load)
        l0.x = fix(l0.x); l0.y = fix(l0.y); l0.z = fix(l0.z); l0.w = fix(l0.w);
        l1.x = fix(l1.x); l1.y = fix(l1.y); l1.z = fix(l1.z); l1.w = fix(l1.w);
        l2.x = fix(l2.x); l2.y = fix(l2.y); l2.z = fix(l2.z); l2.w = fix(l2.w);
        l3.x = fix(l3.x); l3.y = fix(l3.y); l3.z = fix(l3.z); l3.w = fix(l3.w);
        sl4[(ra0 >> 3) + tid] = l0; sl4[(ra0 >> 3) + tid + 64] = l1; sl4[(ra0 >> 3) + tid + 128] = l2; sl4[(ra0 >> 3) + tid + 192] = l3;
        return;
    }
    for (int i = tid; i < R8_C / 4; i += nthreads) {
        const int64_t pos = pos0 + 4 * (int64_t)i;
        uint32_t w = 0;
        if (pos >= 0 && pos + 4 <= seg_end) w = load_u32_unaligned2(d + pos);
        else for (int b = 0; b < 4; b++) { const int64_t pk = pos + b; if (pk >= 0 && pk < seg_end) w |= (uint32_t)d[pk] << (8 * b); }
        sdata32[(ra0 >> 2) + i] = w;
        if (ra0 == 0 && 4 * i < R8_PAD) sdata32[(R8_W >> 2) + i] = w;      // the bytes after the ring's end are those at its start
    }
    for (int i = tid; i < R8_C / 2; i += nthreads) {
        const int64_t pos = pos0 + 2 * (int64_t)i;
        uint32_t w = 0;
        if (pos >= 0 && pos < link_end) w |= lk[pos];
        if (pos + 1 >= 0 && pos + 1 < link_end) w |= (uint32_t)lk[pos + 1] << 16;
        if ((w & 0xFFFFu) == 0) w |= 0xFFFFu;
        if ((w >> 16) == 0) w |= 0xFFFF0000u;
        slink32[(ra0 >> 1) + i] = w;
    }
}

__global__ __launch_bounds__(B2_THREADS) void k_match8(const uint8_t *__restrict__ in, const SegDev *__restrict__ segs,
                                                       const TileDev *__restrict__ stripes, const uint16_t *__restrict__ link,
                                                       MTab mtab, LevelParams P, int fth, int vth, int qkeep, int vkeep, int slice, int lowwater, unsigned long long *dbg) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const TileDev stripe = stripes[blockIdx.x];
    const SegDev seg = segs[stripe.seg];
    const uint16_t *slink = (const uint16_t *)(smem + R8_LINK_OFF);
    volatile int *ctl = (volatile int *)(smem + R8_CTL_OFF);
    int *ctl_a = (int *)(smem + R8_CTL_OFF);
    const uint8_t *d = in + seg.buf_off;
    const uint16_t *lk = link + seg.buf_off;
    uint32_t *__restrict__ mt2 = mtab.m2 + seg.buf_off;
    uint32_t *__restrict__ mtq = mtab.mq + seg.buf_off;
    const int64_t seg_end = seg.look_end;
    const int64_t stripe_end = stripe.start + stripe.len;
    const int64_t origin = stripe.start - R8_H;                 // stream position of ring position 0 (may be negative)
    const int xo_end = R8_H + stripe.len;                       // positions are counted from the origin ("xo") from here on
    const int NC = (xo_end + R8_PAD + R8_C - 1) / R8_C;         // chunks the stripe needs in all
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    {
        const int n0 = NC < R8_NCH ? NC : (int)R8_NCH;
        for (int k = 0; k < n0; k++) r8_stage_chunk(smem, d, lk, origin, k, seg_end, stripe_end, threadIdx.x, B2_THREADS);
        if (threadIdx.x == 0) { ctl[R8_COUNTER] = R8_H; ctl[R8_STAGED] = n0; ctl[R8_LOCK] = 0; }
        if (threadIdx.x < 16) ctl[R8_SLOT0 + threadIdx.x] = R8_H;
    }
    __syncthreads();

    const uint8_t *sdata8 = smem;
    const uint32_t dbase = (uint32_t)(uintptr_t)(lds_u8 *)smem;
    const uint32_t lbase = dbase + (uint32_t)R8_LINK_OFF;
    const uint64_t lanemask_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int SNAPLEFT = P.max_chain - (P.max_chain >> 2);
    const int INF = 0x7FFFFFFF;

    WalkCtx8 A, B;
    A.p = 0; A.pg = 0; A.cl = 0; A.dist = 0; A.maxd = 0; A.best = 2; A.left = 0; A.off = 0; A.cap = MAX_MATCH; A.nice = P.nice; A.pb = 0; A.res2 = 0; A.resq = 0;
    B = A;
    uint64_t qA = 0, vA = 0, dA = 0, qB = 0, vB = 0, dB = 0;
    int wnext = 0, wend = 0;          // the wave's reservation (xo)
    int myslot = R8_H;                // what ctl[R8_SLOT0 + wave] holds: a lower bound of every position this wave still works on
    bool done_all = false;            // the counter passed the stripe's end
    unsigned c_iter = 0, c_starve = 0, c_blocked = 0, c_lockfail = 0, c_sleep = 0, c_staged = 0, c_busy = 0;   // (lab counters, per wave)

    auto wave_min = [&](int v) { for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(v, o); v = t < v ? t : v; } return v; };
    auto retire = [&](WalkCtx8 &C, uint64_t &dm) {
        if (__builtin_amdgcn_inverse_ballot_w64(dm)) { const uint32_t e = mt_pack(C.res2, C.resq); mt2[origin + C.pg] = e; if (e >> 25) mtq[origin + C.pg] = C.resq; }
        dm = 0;
    };
    // start walks on the free lanes of a context with the positions [wnext, lim)
    auto assign = [&](WalkCtx8 &C, uint64_t &q, uint64_t &v, int lim) {
        if (wnext >= lim) return;
        const uint64_t idle = ~(q | v);
        const int ni = __builtin_popcountll(idle);
        if (ni == 0) return;
        const int rank = __builtin_popcountll(idle & lanemask_lt);
        const int rabase = wnext % (int)R8_W;
        bool toverify = false;
        if (__builtin_amdgcn_inverse_ballot_w64(idle) && wnext + rank < lim) {
            const int pxo = wnext + rank;
            int ra = rabase + rank;
            ra = ra >= (int)R8_W ? ra - (int)R8_W : ra;
            C.pg = pxo; C.p = ra;
            const int64_t pos = origin + pxo;
            const int64_t rem64 = seg_end - pos;
            const int rem = rem64 > (int64_t)(1 << 24) ? (1 << 24) : (int)rem64;
            C.res2 = 0; C.resq = 0;
            bool ok = rem >= MIN_MATCH && P.strategy != 2;              // :780, HuffmanOnly :786
            if (ok) {
                const uint32_t l0 = slink[ra];                           // hashHead (:782) as a distance; none = 0xFFFF
                const int64_t s_abs = (int64_t)seg.abs0 + pos;
                const int64_t room64 = s_abs - base_of2(s_abs);         // distance to window index 0 (entries below were clamped by a slide, :450-461)
                const uint32_t room = room64 > (int64_t)(1 << 20) ? (1u << 20) : (uint32_t)room64;
                ok = l0 <= (room < (uint32_t)MAX_DIST ? room : (uint32_t)MAX_DIST);   // strstart - hashHead <= MAX_DIST (:788)
                if (ok) {
                    int c = ra - (int)l0;
                    C.cl = c < 0 ? c + (int)R8_W : c;
                    C.dist = l0;
                    C.maxd = room < (uint32_t)(MAX_DIST - 1) ? room : (uint32_t)(MAX_DIST - 1);   // curMatch > limit (:609)
                    C.cap = rem < MAX_MATCH ? rem : MAX_MATCH;
                    C.nice = rem < P.nice ? rem : P.nice;
                    C.best = 2; C.left = P.max_chain - 1;
                    C.pb = ((uint32_t)sdata8[ra + 2] << 8) | sdata8[ra + 1];
                    C.off = 0;
                    toverify = true;
                }
            }
            if (!ok) mt2[pos] = 0u;
        }
        v |= __ballot(toverify);
        wnext = wnext + ni < lim ? wnext + ni : lim;
    };

    for (;;) {
        retire(A, dA);
        retire(B, dB);
        const int nidle = 128 - __builtin_popcountll(qA | vA) - __builtin_popcountll(qB | vB);
        if (wnext >= wend && !done_all && nidle > 0) {                  // a new reservation
            int base = 0, c0 = 0;
            if (lane == 0) c0 = ctl[R8_COUNTER];
            c0 = __builtin_amdgcn_readfirstlane(c0);
            if (myslot > c0) {        // (a wave with nothing in flight: whatever it takes next is at or above the counter — say so BEFORE taking it)
                myslot = c0;
                if (lane == 0) ctl[R8_SLOT0 + wave] = c0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            }
            if (lane == 0) base = atomicAdd(&ctl_a[R8_COUNTER], slice);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= xo_end) done_all = true;
            else { wnext = base; wend = base + slice < xo_end ? base + slice : xo_end; }
        }
        int st = 0, cnt = 0;
        if (lane == 0) { st = ctl[R8_STAGED]; cnt = ctl[R8_COUNTER]; }
        st = __builtin_amdgcn_readfirstlane(st); cnt = __builtin_amdgcn_readfirstlane(cnt);
        int avail = st == NC ? xo_end : st * (int)R8_C - (int)R8_PAD;
        if (st < NC && (avail - cnt < lowwater || (wnext < wend && wnext >= avail))) {
            // keep the ring ahead of the counter: one wave at a time stages the next chunk, over the oldest one
            int got = 0;
            if (lane == 0) got = atomicCAS(&ctl_a[R8_LOCK], 0, 1) == 0;
            got = __builtin_amdgcn_readfirstlane(got);
            if (!got) c_lockfail++;
            if (got) {
                int st2 = 0, c1 = 0;
                if (lane == 0) { st2 = ctl[R8_STAGED]; c1 = ctl[R8_COUNTER]; }           // the counter BEFORE the slots (see the reservation above)
                st2 = __builtin_amdgcn_readfirstlane(st2); c1 = __builtin_amdgcn_readfirstlane(c1);
                if (st2 < NC) {
                    int mn = lane < 16 ? ctl[R8_SLOT0 + lane] : INF;
                    mn = wave_min(mn);
                    mn = mn < c1 ? mn : c1;
                    // chunk k goes over chunk k - R8_NCH: no walk may reach below position (k - R8_NCH + 1) * R8_C any more
                    int k = st2;
                    while (k < NC && k < st2 + 3 && mn >= (k - (int)R8_NCH + 1) * (int)R8_C + (int)B_HIST) {
                        r8_stage_chunk(smem, d, lk, origin, k, seg_end, stripe_end, lane, 64);
                        k++;
                    }
                    if (k > st2) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) ctl[R8_STAGED] = k;
                        c_staged += (unsigned)(k - st2);
                    } else c_blocked++;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) { atomicExch(&ctl_a[R8_LOCK], 0); st = ctl[R8_STAGED]; }
                st = __builtin_amdgcn_readfirstlane(st);
                avail = st == NC ? xo_end : st * (int)R8_C - (int)R8_PAD;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int lim = wend < avail ? wend : avail;
        assign(A, qA, vA, lim);
        assign(B, qB, vB, lim);
        {   // publish the lowest position this wave still works on
            const bool actA = __builtin_amdgcn_inverse_ballot_w64(qA | vA), actB = __builtin_amdgcn_inverse_ballot_w64(qB | vB);
            int mn = actA ? A.pg : INF;
            mn = actB && B.pg < mn ? B.pg : mn;
            mn = wave_min(mn);
            if (wnext < wend && wnext < mn) mn = wnext;
            if (mn != myslot) { myslot = mn; if (lane == 0) ctl[R8_SLOT0 + wave] = mn; }
        }
        const uint32_t busy = (uint32_t)(__builtin_popcountll(qA | vA) + __builtin_popcountll(qB | vB));
        if (busy == 0) {
            if (done_all && wnext >= wend) break;
            c_sleep++;
            __builtin_amdgcn_s_sleep(4);
            continue;
        }
        // run until `fth` contexts are free again; without positions to give them (the ring is not ahead, or the stripe is handed out)
        // until 16 more walks are over, resp. all of them
        const bool handed_out = done_all && wnext >= wend;
        const bool starving = wnext >= lim && !handed_out;
        c_iter++; c_busy += busy; if (starving) c_starve++;
        const uint32_t busy_exit = (uint32_t)__builtin_amdgcn_readfirstlane((int)(handed_out ? 0u : (starving ? (busy > 16u ? busy - 16u : 0u) : (uint32_t)(128 - fth))));
        uint32_t t0A, t1A, t2A, t3A, t4A, t5A, t6A, t7A, t0B, t1B, t2B, t3B, t4B, t5B, t6B, t7B;
        uint64_t mA, mB, sc, cm, sv;
        uint32_t n0, n1, n2;
        asm volatile(
            SZL_ENGINE_TEXT(SZL8_Q_FINISH, SZL8_Q_FINISH_LAST, SZL8_V_COMPLETE, "", "", "")
            : [pA] "+&v"(A.p), [clA] "+&v"(A.cl), [bestA] "+&v"(A.best), [leftA] "+&v"(A.left), [distA] "+&v"(A.dist), [offA] "+&v"(A.off), [pbA] "+&v"(A.pb),
              [res2A] "+&v"(A.res2), [resqA] "+&v"(A.resq),
              [pB] "+&v"(B.p), [clB] "+&v"(B.cl), [bestB] "+&v"(B.best), [leftB] "+&v"(B.left), [distB] "+&v"(B.dist), [offB] "+&v"(B.off), [pbB] "+&v"(B.pb),
              [res2B] "+&v"(B.res2), [resqB] "+&v"(B.resq),
              [qA] "+&s"(qA), [vA] "+&s"(vA), [dA] "+&s"(dA), [qB] "+&s"(qB), [vB] "+&s"(vB), [dB] "+&s"(dB),
              [t0A] "=&v"(t0A), [t1A] "=&v"(t1A), [t2A] "=&v"(t2A), [t3A] "=&v"(t3A), [t4A] "=&v"(t4A), [t5A] "=&v"(t5A), [t6A] "=&v"(t6A), [t7A] "=&v"(t7A),
              [t0B] "=&v"(t0B), [t1B] "=&v"(t1B), [t2B] "=&v"(t2B), [t3B] "=&v"(t3B), [t4B] "=&v"(t4B), [t5B] "=&v"(t5B), [t6B] "=&v"(t6B), [t7B] "=&v"(t7B),
              [mA] "=&s"(mA), [mB] "=&s"(mB), [sc] "=&s"(sc), [cm] "=&s"(cm), [sv] "=&s"(sv), [n0] "=&s"(n0), [n1] "=&s"(n1), [n2] "=&s"(n2)
            : [maxdA] "v"(A.maxd), [capA] "v"(A.cap), [niceA] "v"(A.nice), [maxdB] "v"(B.maxd), [capB] "v"(B.cap), [niceB] "v"(B.nice),
              [lbase] "s"(lbase), [dbase] "s"(dbase), [pbase] "s"(dbase), [dbm1] "s"(dbase - 1u), [pbm1] "s"(dbase - 1u), [snap] "s"(SNAPLEFT),
              [bexit] "s"(busy_exit), [ringw] "s"((uint32_t)R8_W), [vth] "s"(vth), [qkeep] "s"(qkeep), [vkeep] "s"(vkeep)
            : "vcc", "scc", "memory");
    }
    if (dbg && lane == 0) {
        atomicAdd(dbg + 22, (unsigned long long)c_iter); atomicAdd(dbg + 23, (unsigned long long)c_starve); atomicAdd(dbg + 24, (unsigned long long)c_blocked);
        atomicAdd(dbg + 25, (unsigned long long)c_lockfail); atomicAdd(dbg + 26, (unsigned long long)c_sleep); atomicAdd(dbg + 27, (unsigned long long)c_staged);
        atomicAdd(dbg + 28, (unsigned long long)c_busy);
    }
}

#endif   // SZL_LAB

#if SZL_LAB
static bool lds_attr_needed2(std::atomic<uint64_t> &mask, uint64_t &bit) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    bit = 1ull << (dev & 63);
    return (mask.load(std::memory_order_acquire) & bit) == 0;
}
#endif

hipError_t launch_match9(const uint8_t *in, const SegDev *segs, const TileDev *tiles, int ntiles, const uint16_t *link, MTab mtab, LevelParams P,
                         unsigned long long *dbg, hipStream_t st);

hipError_t launch_match2(const uint8_t *in, const SegDev *segs, const TileDev *tiles, int ntiles, const uint16_t *link,
                         MTab mtab, LevelParams P, unsigned long long *dbg, hipStream_t st) {
#if !SZL_LAB
    return launch_match9(in, segs, tiles, ntiles, link, mtab, P, dbg, st);   // the all-assembly engine of szl_kernels_match9.hip
#else
    // SZL_B9 (laboratory): 1 = k_match9 (same tiles, same tables); 0 = k_match4
    if (SZL_LABKNOB("SZL_B9", SZL_B9_DEFAULT) != 0) return launch_match9(in, segs, tiles, ntiles, link, mtab, P, dbg, st);
    static std::atomic<uint64_t> attr_mask{0};
    uint64_t attr_bit = 0;
    const bool want_dbg = knob("SZL_DEBUG", 0) != 0;
    // thresholds count CONTEXTS (two per lane, 128 per wavefront)
    int fth = SZL_LABKNOB("SZL_FTH2", 32), vth = SZL_LABKNOB("SZL_VTH2", 2), qkeep = SZL_LABKNOB("SZL_QKEEP", 64), vkeep = SZL_LABKNOB("SZL_VKEEP", 2); // swept: profiles/r02/lab_s6_k_match4_sweep.log, lab_s15_two_byte_filter.log, lab_s37_slice_and_thresholds.log
    auto attr = [&](const void *f) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, B2_LDS_BYTES); };
    if (lds_attr_needed2(attr_mask, attr_bit)) {
        hipError_t e = attr((const void *)k_match4<false>);
        if (e == hipSuccess) e = attr((const void *)k_match4<true>);
        if (e != hipSuccess) return e;
        attr_mask.fetch_or(attr_bit, std::memory_order_release);
    }
    if (want_dbg && SZL_LABKNOB("SZL_B_EXP", 0) == 1) fth = -1;      // lab: stage-only timing experiment (results are garbage)
    else if (fth < 1) fth = 1;
    if (fth > 128) fth = 128;
    if (SZL_LABKNOB("SZL_B_CHAIN", 0) > 0) P.max_chain = SZL_LABKNOB("SZL_B_CHAIN", 0); // lab: shorter chains (results differ from the reference)
    if (vth < 1) vth = 1;
    if (qkeep < 1) qkeep = 1;
    if (vkeep < 1) vkeep = 1;
    int slice = SZL_LABKNOB("SZL_SLICE", 128);   // tile positions a wavefront takes from the tile counter at a time (512: 63.5, 128: 61.6 ms per GiB — a shorter tail per tile)
    slice = slice < 64 ? 64 : (slice > 4096 ? 4096 : slice);
    if (ntiles > 0) {
        const dim3 g(ntiles), b(B2_THREADS);
        if (want_dbg) hipLaunchKernelGGL((k_match4<true>), g, b, B2_LDS_BYTES, st, in, segs, tiles, link, mtab, P, dbg, fth, vth, qkeep, vkeep, slice);
        else hipLaunchKernelGGL((k_match4<false>), g, b, B2_LDS_BYTES, st, in, segs, tiles, link, mtab, P, dbg, fth, vth, qkeep, vkeep, slice);
    }
    return hipGetLastError();
#endif
}

#if SZL_LAB
// stripes: TileDev entries of any length (the engine cuts them)
hipError_t launch_match_ring(const uint8_t *in, const SegDev *segs, const TileDev *stripes, int nstripes, const uint16_t *link, MTab mtab, LevelParams P,
                             unsigned long long *dbg, hipStream_t st) {
    static std::atomic<uint64_t> attr_mask{0};
    uint64_t attr_bit = 0;
    int fth = SZL_LABKNOB("SZL_FTH2", 32), vth = SZL_LABKNOB("SZL_VTH2", 2), qkeep = SZL_LABKNOB("SZL_QKEEP", 64), vkeep = SZL_LABKNOB("SZL_VKEEP", 2), slice = SZL_LABKNOB("SZL_SLICE", 128);
    if (lds_attr_needed2(attr_mask, attr_bit)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_match8, hipFuncAttributeMaxDynamicSharedMemorySize, R8_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_mask.fetch_or(attr_bit, std::memory_order_release);
    }
    fth = fth < 1 ? 1 : (fth > 128 ? 128 : fth); vth = vth < 1 ? 1 : vth; qkeep = qkeep < 1 ? 1 : qkeep; vkeep = vkeep < 1 ? 1 : vkeep;
    slice = slice < 64 ? 64 : (slice > 1024 ? 1024 : slice);
    int lowwater = SZL_LABKNOB("SZL_LOWWATER", 6144);
    lowwater = lowwater < 512 ? 512 : (lowwater > 16384 ? 16384 : lowwater);
    // (fewer than 16 waves per workgroup — the timing model's suggestion, DESIGN §8 of round 2 — measured: 12 / 10 / 8 / 6 waves take 57.6 / 62.3 /
    // 68.3 / 84.2 ms per GiB against 54.9 with 16, profiles/r03/lab_r3a_ring_waves_256.log; the variant was removed)
    unsigned long long *dbg8 = knob("SZL_DEBUG", 0) ? dbg : nullptr;
    if (nstripes > 0)
        hipLaunchKernelGGL(k_match8, dim3(nstripes), dim3(B2_THREADS), R8_LDS_BYTES, st, in, segs, stripes, link, mtab, P, fth, vth, qkeep, vkeep, slice, lowwater, dbg8);
    return hipGetLastError();
}

#endif

int match2_tile() { return B2_TILE; }

} // namespace szl
