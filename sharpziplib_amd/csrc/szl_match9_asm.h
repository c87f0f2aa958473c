// szl_match9_asm.h — instruction text of k_match9, stage B's full search (FindLongestMatch for every position of a tile,
// C/DeflaterEngine.cs:474-612): the WHOLE main loop — retire, fetch, chain walk, compare, bookkeeping — is hand-written gfx950
// assembly.  Pure preprocessor text (no HIP) so that tools/wavesim.py can run the same text on the CPU (tools/sim_match9.py).
//
// Why (round-3 PMC, profiles/r03/pmc_sq_1gib.json): k_match4 is bound by VALU issue — 18.1 wave-instructions per position, of
// which 6.9 are the chain walk (7 per step), 3.7 the compare, 2.2 its completion and 3.3 the compiler-generated fetch (incl. ~40
// register moves around every call of the inline-assembly engine).  What changed against k_match4's engine:
//   * QUICK (one chain step, :502-507,:609) keeps cb = candidate + best_len instead of the candidate: the address of the filter
//     bytes IS the state variable and the link address is 2*cb + kk with kk = -2*best_len in a register: 6 VALU per step, not 7
//     (5 with the one-byte filter, SZL9_NQ = 2);
//   * a QUICK step hops BEFORE it tests the filter, so a lane that leaves for a compare already stands on the NEXT candidate (the
//     one to compare is cb + hop - best_len): completing a compare needs no prev[] read, no hop and no end-of-chain arithmetic —
//     a lane whose compare did not improve on best_len goes back to the walk untouched;
//   * the first 16 bytes of the position are kept in four registers: the first VERIFY step of a candidate compares 16 bytes
//     against them (5 aligned dword reads instead of 12); 88 % of the compares on text end there (63 % within 8 bytes), and every
//     compare that does not costs a pass of the generic loop AND a second completion pass at a few lanes each;
//   * FETCH is assembly: no register shuffles around the engine, one context per visit (the one with more free lanes), window
//     base / lookahead clamps chosen by scalar code per slice (they are wave-uniform except in the slice that holds a slide);
//   * base addresses are instruction offsets (the window sits at a fixed LDS address).
// Semantics are those of k_match4 (same tables, bit for bit): M2 = walk entered with matchLen 2 and the full budget, Mq = its
// state after max_chain >> 2 candidates.
//
// LDS layout (bytes): [0,4) tile counter, [SZL9_D, SZL9_D + DATA) window bytes, [SZL9_LB, ...) links (u16, "none" = 0xFFFF).
// An LDS *index* x is a window position (0 = first history byte); byte x lives at x + SZL9_D, its link at 2x + SZL9_LB.
#pragma once

#define SZL9_XSTR(x) #x
#define SZL9_STR(x) SZL9_XSTR(x)

// The text has two FORMS, picked where SZL9_TEXT is expanded by the value SZL9_V has there (every macro below is expanded only then):
//   SZL9_V 0  the filter bytes are the reference's, scan_end1 and scan_end (C/DeflaterEngine.cs:512-515);
//   SZL9_V 1  the filter's FIRST byte follows the lane's last failed compare (SZL9_Q_ISSUE_1): the form for tiles whose chains are dense.
// k_match9 holds both and a tile takes one (szl_kernels_match9.hip: the share of short prev[] hops in the tile's window decides).
#ifndef SZL9_V
#define SZL9_V 0
#endif
#define SZL9_CAT_(a, b) a##b
#define SZL9_CAT(a, b) SZL9_CAT_(a, b)
#define SZL9_FORM(name) SZL9_CAT(name, SZL9_V)
#ifndef SZL9_NQ
#define SZL9_NQ 3              // LDS reads per QUICK step: 3 = two filter bytes (scan_end1, scan_end), 2 = scan_end only
#endif

// numbers the text needs as literals (checked against the enums by static_assert in the kernel)
#define SZL9_D 16
#define SZL9_DM1 15
#define SZL9_D1 17
#define SZL9_D2 18
#define SZL9_D4 20
#define SZL9_D8 24
#define SZL9_D12 28
#define SZL9_D16 32
#define SZL9_LB 54304          // SZL9_D + B2_DATA_BYTES
#define SZL9_BHIST 32512

// ---- QUICK: one chain step of context X ------------------------------------------------------------------------------------
// State of a walking lane: cb = candidate + best_len (LDS index of the candidate's byte at best_len), hop = prev[] distance read
// for it, left = candidates that may still follow, pb = the position's filter byte(s), mincb = first index no candidate may
// fall below (+ best_len).  A step reads the candidate's hop and filter byte(s), moves on to the next candidate, and leaves the
// loop (exec) when the budget is spent (:609 --chainLength), the next candidate is out of the window (:609 curMatch > limit) or
// the filter bytes match (:505-506; the candidate to compare is then cb + hop - best_len).
#if SZL9_NQ == 3
// Form 1: the first filter byte is the candidate's byte at offset best_len + kd: kd = -1 (scan_end1, the reference's) until a compare of
// this walk fails without improving best_len; from then on the offset at which it failed (COMPLETE).  Any byte in front of best_len is
// a valid filter — a candidate that differs there cannot beat best_len — and on data whose lines differ from the position's in one
// field the next candidates of the chain fail where the last one did: 4.6 of 5.7 compares per position at level 9 on logs do not
// improve, 70 % of them are caught by this byte (tools/lab/walkstat.c).  One VALU instruction more per chain step (+1 % on text, where
// it catches a third of 14 %): a tile takes this form where its chains are dense.
#define SZL9_Q_ISSUE_1(X) \
    "v_lshl_add_u32 %[t0" #X "], %[cb" #X "], 1, %[kk" #X "]\n\t" \
    "v_add_u32 %[t3" #X "], %[cb" #X "], %[kd" #X "]\n\t" \
    "ds_read_u16 %[hop" #X "], %[t0" #X "] offset:" SZL9_STR(SZL9_LB) "\n\t"     /* prev[] hop of the candidate */ \
    "ds_read_u8 %[t2" #X "], %[t3" #X "] offset:" SZL9_STR(SZL9_D) "\n\t"        /* candidate[best_len + kd] */ \
    "ds_read_u8 %[t1" #X "], %[cb" #X "] offset:" SZL9_STR(SZL9_D) "\n\t"        /* candidate[best_len] */
#define SZL9_Q_ISSUE(X) SZL9_FORM(SZL9_Q_ISSUE_)(X)
#define SZL9_Q_ISSUE_0(X) \
    "v_lshl_add_u32 %[t0" #X "], %[cb" #X "], 1, %[kk" #X "]\n\t" \
    "ds_read_u16 %[hop" #X "], %[t0" #X "] offset:" SZL9_STR(SZL9_LB) "\n\t"     /* prev[] hop of the candidate */ \
    "ds_read_u8 %[t2" #X "], %[cb" #X "] offset:" SZL9_STR(SZL9_DM1) "\n\t"      /* candidate[best_len - 1] */ \
    "ds_read_u8 %[t1" #X "], %[cb" #X "] offset:" SZL9_STR(SZL9_D) "\n\t"        /* candidate[best_len] */
/* (ds_read_u8_d16_hi into the register the first byte is in flight to would save this instruction — measured on the device it does
 * not keep the low half: gfx950 runs with SRAM ECC, where a d16 load writes the whole register; profiles/r04/c1_lab.log) */
#define SZL9_Q_COMBINE(X) "v_lshl_or_b32 %[t1" #X "], %[t1" #X "], 16, %[t2" #X "]\n\t"
#else
#define SZL9_Q_ISSUE(X) \
    "v_lshl_add_u32 %[t0" #X "], %[cb" #X "], 1, %[kk" #X "]\n\t" \
    "ds_read_u16 %[hop" #X "], %[t0" #X "] offset:" SZL9_STR(SZL9_LB) "\n\t" \
    "ds_read_u8 %[t1" #X "], %[cb" #X "] offset:" SZL9_STR(SZL9_D) "\n\t"
#define SZL9_Q_COMBINE(X)
#endif
#define SZL9_Q_FINISH_(X, TAIL) \
    SZL9_Q_COMBINE(X) \
    "v_sub_u32 %[cb" #X "], %[cb" #X "], %[hop" #X "]\n\t" \
    "v_subrev_co_u32 %[left" #X "], vcc, 1, %[left" #X "]\n\t"                     /* --chainLength != 0 */ \
    "s_andn2_b64 exec, exec, vcc\n\t" \
    "v_cmpx_ge_i32 vcc, %[cb" #X "], %[mincb" #X "]\n\t"                           /* curMatch > limit, window index >= 1 (:609) */ \
    "v_cmpx_ne_u32 vcc, %[pb" #X "], %[t1" #X "]\n\t" \
    TAIL
#define SZL9_Q_FINISH(X) SZL9_Q_FINISH_(X, "s_mov_b64 %[m" #X "], exec\n\t")
#define SZL9_Q_FINISH_LAST(X) SZL9_Q_FINISH_(X, "s_mov_b64 %[m" #X "], exec\n\t")
// leavers of context X: the filter byte(s) read last say whether the candidate they stood on is to be compared; the others'
// walks are over (only the budget or the window limit takes a lane out before the filter test)
#define SZL9_Q_CLASSIFY(X) \
    "s_andn2_b64 exec, %[q" #X "], %[m" #X "]\n\t" \
    "v_cmp_eq_u32 vcc, %[pb" #X "], %[t1" #X "]\n\t" \
    "s_or_b64 %[v" #X "], %[v" #X "], vcc\n\t" \
    "s_andn2_b64 %[sc], exec, vcc\n\t" \
    "s_or_b64 %[d" #X "], %[d" #X "], %[sc]\n\t" \
    "s_mov_b64 %[q" #X "], %[m" #X "]\n\t"

// ---- VERIFY, first step of a candidate: 16 bytes of the candidate against the position's first 16 bytes (registers p0-p3) ----
#define SZL9_VF_ISSUE(X) \
    "v_add_u32 %[t0" #X "], %[cb" #X "], %[hop" #X "]\n\t" \
    "v_sub_u32 %[t0" #X "], %[t0" #X "], %[best" #X "]\n\t"                       /* the candidate (LDS index) */ \
    "v_and_b32 %[t5" #X "], -4, %[t0" #X "]\n\t" \
    "ds_read_b32 %[t1" #X "], %[t5" #X "] offset:" SZL9_STR(SZL9_D) "\n\t" \
    "ds_read_b32 %[t2" #X "], %[t5" #X "] offset:" SZL9_STR(SZL9_D4) "\n\t" \
    "ds_read_b32 %[t3" #X "], %[t5" #X "] offset:" SZL9_STR(SZL9_D8) "\n\t" \
    "ds_read_b32 %[t4" #X "], %[t5" #X "] offset:" SZL9_STR(SZL9_D12) "\n\t" \
    "ds_read_b32 %[t5" #X "], %[t5" #X "] offset:" SZL9_STR(SZL9_D16) "\n\t"
#define SZL9_VF_FINISH(X) \
    "v_alignbyte_b32 %[t1" #X "], %[t2" #X "], %[t1" #X "], %[t0" #X "]\n\t" \
    "v_alignbyte_b32 %[t2" #X "], %[t3" #X "], %[t2" #X "], %[t0" #X "]\n\t" \
    "v_alignbyte_b32 %[t3" #X "], %[t4" #X "], %[t3" #X "], %[t0" #X "]\n\t" \
    "v_alignbyte_b32 %[t4" #X "], %[t5" #X "], %[t4" #X "], %[t0" #X "]\n\t" \
    "v_xor_b32 %[t1" #X "], %[t1" #X "], %[p0" #X "]\n\t" \
    "v_xor_b32 %[t2" #X "], %[t2" #X "], %[p1" #X "]\n\t" \
    "v_xor_b32 %[t3" #X "], %[t3" #X "], %[p2" #X "]\n\t" \
    "v_xor_b32 %[t4" #X "], %[t4" #X "], %[p3" #X "]\n\t" \
    "v_ffbl_b32 %[t1" #X "], %[t1" #X "]\n\t" \
    "v_ffbl_b32 %[t2" #X "], %[t2" #X "]\n\t" \
    "v_ffbl_b32 %[t3" #X "], %[t3" #X "]\n\t" \
    "v_ffbl_b32 %[t4" #X "], %[t4" #X "]\n\t" \
    "v_or_b32 %[t2" #X "], 32, %[t2" #X "]\n\t" \
    "v_or_b32 %[t3" #X "], 64, %[t3" #X "]\n\t" \
    "v_or_b32 %[t4" #X "], 0x60, %[t4" #X "]\n\t" \
    "v_min3_u32 %[t1" #X "], %[t1" #X "], %[t2" #X "], %[t3" #X "]\n\t" \
    "v_min_u32 %[t1" #X "], %[t1" #X "], %[t4" #X "]\n\t"                         /* first differing bit of the 128 */ \
    "v_lshrrev_b32 %[t1" #X "], 3, %[t1" #X "]\n\t" \
    "v_min_u32 %[off" #X "], 16, %[t1" #X "]\n\t" \
    "v_cmpx_eq_u32 vcc, 16, %[off" #X "]\n\t" \
    "v_cmpx_lt_i32 vcc, %[off" #X "], %[cap" #X "]\n\t" \
    "s_mov_b64 %[m" #X "], exec\n\t"
// ---- VERIFY, later steps (off >= 16): 8 bytes of each side from three ALIGNED dwords + v_alignbyte -------------------------
#define SZL9_V_ISSUE(X) \
    "v_add3_u32 %[t0" #X "], %[cb" #X "], %[hop" #X "], %[off" #X "]\n\t" \
    "v_add_u32 %[t1" #X "], %[pl" #X "], %[off" #X "]\n\t" \
    "v_sub_u32 %[t0" #X "], %[t0" #X "], %[best" #X "]\n\t" \
    "v_and_b32 %[t5" #X "], -4, %[t1" #X "]\n\t" \
    "v_and_b32 %[t4" #X "], -4, %[t0" #X "]\n\t" \
    "ds_read_b32 %[t2" #X "], %[t4" #X "] offset:" SZL9_STR(SZL9_D) "\n\t" \
    "ds_read_b32 %[t3" #X "], %[t4" #X "] offset:" SZL9_STR(SZL9_D4) "\n\t" \
    "ds_read_b32 %[t4" #X "], %[t4" #X "] offset:" SZL9_STR(SZL9_D8) "\n\t" \
    "ds_read_b32 %[t6" #X "], %[t5" #X "] offset:" SZL9_STR(SZL9_D) "\n\t" \
    "ds_read_b32 %[t7" #X "], %[t5" #X "] offset:" SZL9_STR(SZL9_D4) "\n\t" \
    "ds_read_b32 %[t5" #X "], %[t5" #X "] offset:" SZL9_STR(SZL9_D8) "\n\t"
#define SZL9_V_FINISH(X) \
    "v_alignbyte_b32 %[t2" #X "], %[t3" #X "], %[t2" #X "], %[t0" #X "]\n\t" \
    "v_alignbyte_b32 %[t3" #X "], %[t4" #X "], %[t3" #X "], %[t0" #X "]\n\t" \
    "v_alignbyte_b32 %[t6" #X "], %[t7" #X "], %[t6" #X "], %[t1" #X "]\n\t" \
    "v_alignbyte_b32 %[t7" #X "], %[t5" #X "], %[t7" #X "], %[t1" #X "]\n\t" \
    "v_xor_b32 %[t2" #X "], %[t2" #X "], %[t6" #X "]\n\t" \
    "v_xor_b32 %[t3" #X "], %[t3" #X "], %[t7" #X "]\n\t" \
    "v_ffbl_b32 %[t2" #X "], %[t2" #X "]\n\t" \
    "v_ffbl_b32 %[t3" #X "], %[t3" #X "]\n\t" \
    "v_or_b32 %[t3" #X "], 32, %[t3" #X "]\n\t" \
    "v_min_u32 %[t2" #X "], %[t2" #X "], %[t3" #X "]\n\t" \
    "v_lshrrev_b32 %[t2" #X "], 3, %[t2" #X "]\n\t" \
    "v_min_u32 %[t2" #X "], 8, %[t2" #X "]\n\t" \
    "v_add_u32 %[off" #X "], %[off" #X "], %[t2" #X "]\n\t" \
    "v_cmpx_eq_u32 vcc, 8, %[t2" #X "]\n\t" \
    "v_cmpx_lt_i32 vcc, %[off" #X "], %[cap" #X "]\n\t" \
    "s_mov_b64 %[m" #X "], exec\n\t"
// ---- COMPLETE (:593-609) for the lanes in c (their compare is over; off = bytes that matched).  Only a lane whose compare beat
// best_len has anything to do: record the match and re-base cb, mincb, kk and the filter bytes on the new best_len.  The lane
// already stands on the next candidate; whether there is one was settled by the QUICK step that sent it here (left == -1: budget
// spent; cb < mincb: out of the window).
// (the body of COMPLETE for the lanes whose compare beat best_len; t2 = the new length; leaves sc = lanes that reached niceLength)
#define SZL9_IMPROVE(X) \
    "v_add_u32 %[t1" #X "], %[cb" #X "], %[hop" #X "]\n\t" \
    "v_sub_u32 %[t3" #X "], %[t2" #X "], %[best" #X "]\n\t"                        /* growth of best_len */ \
    "v_sub_u32 %[t1" #X "], %[t1" #X "], %[best" #X "]\n\t"                        /* the candidate compared */ \
    "v_add_u32 %[cb" #X "], %[cb" #X "], %[t3" #X "]\n\t" \
    "v_sub_u32 %[t1" #X "], %[pl" #X "], %[t1" #X "]\n\t"                          /* distance */ \
    "v_add_u32 %[mincb" #X "], %[mincb" #X "], %[t3" #X "]\n\t" \
    "v_lshl_or_b32 %[res2" #X "], %[t1" #X "], 16, %[t2" #X "]\n\t" \
    "v_add_u32 %[t1" #X "], %[pl" #X "], %[t2" #X "]\n\t" \
    "ds_read_u8 %[t3" #X "], %[t1" #X "] offset:" SZL9_STR(SZL9_DM1) "\n\t"       /* scan_end1 / scan_end for the new best_len */ \
    "ds_read_u8 %[t1" #X "], %[t1" #X "] offset:" SZL9_STR(SZL9_D) "\n\t" \
    "v_cmp_ge_i32 vcc, %[left" #X "], %[snapm1]\n\t"                                /* seen by the quarter-budget walk too (:495); left is already one down */ \
    "v_mov_b32 %[best" #X "], %[t2" #X "]\n\t" \
    "v_cndmask_b32 %[resq" #X "], %[resq" #X "], %[res2" #X "], vcc\n\t" \
    "v_mul_i32_i24 %[kk" #X "], -2, %[t2" #X "]\n\t" \
    "v_cmp_ge_i32 %[sc], %[t2" #X "], %[nice" #X "]\n\t"                           /* >= niceLength: stop (:603) */ \
    "s_waitcnt lgkmcnt(0)\n\t" \
    SZL9_PB_SET(X) \
    SZL9_KD_RESET(X)
#define SZL9_KD_RESET_0(X)
#define SZL9_KD_RESET_1(X) "v_mov_b32 %[kd" #X "], -1\n\t"
/* lanes in exec: their compare did not improve best_len; off = where it failed (t2 = min(off, cap) of COMPLETE).  The first filter byte
 * moves there when that is in front of best_len - 1 (else it stays where it is: nothing learnt) */
#define SZL9_KD_SET_1(X) \
    "v_sub_u32 %[t1" #X "], %[t2" #X "], %[best" #X "]\n\t"                        /* off - best_len */ \
    "v_cmpx_gt_i32 vcc, -1, %[t1" #X "]\n\t" \
    "s_cbranch_execz 36f\n\t" \
    "v_add_u32 %[t3" #X "], %[pl" #X "], %[t2" #X "]\n\t" \
    "ds_read_u8 %[t3" #X "], %[t3" #X "] offset:" SZL9_STR(SZL9_D) "\n\t"       /* the position's byte there */ \
    "v_mov_b32 %[kd" #X "], %[t1" #X "]\n\t" \
    "v_and_b32 %[pb" #X "], 0xffff0000, %[pb" #X "]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_or_b32 %[pb" #X "], %[pb" #X "], %[t3" #X "]\n" \
    "36:\n\t"
#define SZL9_KD_RESET(X) SZL9_FORM(SZL9_KD_RESET_)(X)
/* (sc = improved lanes that reached niceLength; the improved set itself is gone by here: a lane improved iff its best_len == t2 now) */
#define SZL9_COMPLETE_LEARN_1(X) \
    "s_mov_b64 exec, %[c" #X "]\n\t" \
    "v_cmpx_ne_u32 vcc, %[t2" #X "], %[best" #X "]\n\t" \
    "s_cbranch_execz 36f\n\t" \
    SZL9_KD_SET_1(X)
#define SZL9_COMPLETE_LEARN_0(X)
#define SZL9_COMPLETE_LEARN(X) SZL9_FORM(SZL9_COMPLETE_LEARN_)(X)
#define SZL9_COMPLETE(X) \
    "s_mov_b64 exec, %[c" #X "]\n\t" \
    "s_cbranch_execz 39f\n\t" \
    "v_min_i32 %[t2" #X "], %[off" #X "], %[cap" #X "]\n\t"                        /* L */ \
    "v_cmp_gt_i32 %[sc], %[t2" #X "], %[best" #X "]\n\t"                           /* strictly longer: the new best */ \
    "v_cmp_lt_i32 vcc, 2, %[t2" #X "]\n\t"                                         /* ... and a match at all (:611; the walk starts at 1, see FETCH) */ \
    "s_and_b64 %[sc], %[sc], vcc\n\t" \
    "s_mov_b64 exec, %[sc]\n\t" \
    "s_cbranch_execz 38f\n\t" \
    SZL9_IMPROVE(X) "\n" \
    "38:\n\t"                                                                        /* (nobody improved: sc = 0, exec = 0) */ \
    SZL9_COMPLETE_LEARN(X) \
    "s_mov_b64 exec, %[c" #X "]\n\t" \
    "v_cmp_lt_i32 vcc, %[cb" #X "], %[mincb" #X "]\n\t"                            /* the next candidate is out of the window */ \
    "s_or_b64 %[sc], %[sc], vcc\n\t" \
    "v_cmp_gt_i32 vcc, 0, %[left" #X "]\n\t"                                        /* the budget was spent on this candidate */ \
    "s_or_b64 %[sc], %[sc], vcc\n\t"                                                /* sc = lanes whose walk ends here */ \
    "s_or_b64 %[d" #X "], %[d" #X "], %[sc]\n\t" \
    "s_andn2_b64 %[sc], %[c" #X "], %[sc]\n\t" \
    "s_or_b64 %[q" #X "], %[q" #X "], %[sc]\n" \
    "39:\n\t"
#if SZL9_NQ == 3
#define SZL9_PB_SET(X) "v_lshl_or_b32 %[pb" #X "], %[t1" #X "], 16, %[t3" #X "]\n\t"
#else
#define SZL9_PB_SET(X) "v_mov_b32 %[pb" #X "], %[t1" #X "]\n\t"
#endif
// after the first step: the lanes whose compare is over are to be completed (c), those still comparing go on in w
#define SZL9_AFTER_VF(X) \
    "s_andn2_b64 %[c" #X "], %[v" #X "], %[m" #X "]\n\t" \
    "s_or_b64 %[w" #X "], %[w" #X "], %[m" #X "]\n\t" \
    "s_mov_b64 %[v" #X "], 0\n\t"
// one later step of context X (not pipelined with the other context: these are rare)
#define SZL9_W_STEP(X) \
    "s_mov_b64 exec, %[w" #X "]\n\t" \
    "s_cbranch_execz 37f\n\t" \
    SZL9_V_ISSUE(X) \
    "s_waitcnt lgkmcnt(0)\n\t" \
    SZL9_V_FINISH(X) \
    "s_andn2_b64 %[sc], %[w" #X "], %[m" #X "]\n\t" \
    "s_or_b64 %[c" #X "], %[c" #X "], %[sc]\n\t" \
    "s_mov_b64 %[w" #X "], %[m" #X "]\n" \
    "37:\n\t"

// ---- RETIRE: store the finished walks of context X (one packed word per position, szl_internal.h mt_pack) ---------------------
#define SZL9_RETIRE(X) \
    "s_mov_b64 exec, %[d" #X "]\n\t" \
    "s_cbranch_execz 21f\n\t" \
    "v_lshrrev_b32 %[t0" #X "], 16, %[res2" #X "]\n\t" \
    "v_and_b32 %[t1" #X "], 0x1ff, %[res2" #X "]\n\t" \
    "v_cmp_ne_u32 vcc, %[res2" #X "], %[resq" #X "]\n\t" \
    "v_cmp_ne_u32 %[sc], 0, %[resq" #X "]\n\t" \
    "v_lshl_or_b32 %[t0" #X "], %[t0" #X "], 9, %[t1" #X "]\n\t" \
    "v_cndmask_b32 %[t1" #X "], 0, 1, vcc\n\t"                                     /* Mq differs from M2 */ \
    "s_and_b64 %[sc], %[sc], vcc\n\t"                                               /* ... and is not empty: code 2 */ \
    "v_lshlrev_b32 %[t2" #X "], 2, %[pl" #X "]\n\t" \
    "v_cndmask_b32 %[t3" #X "], 0, 1, %[sc]\n\t" \
    "v_add_u32 %[t1" #X "], %[t1" #X "], %[t3" #X "]\n\t" \
    "v_lshl_or_b32 %[t0" #X "], %[t1" #X "], 24, %[t0" #X "]\n\t" \
    "global_store_dword %[t2" #X "], %[t0" #X "], %[mt2b]\n\t" \
    "s_and_b64 exec, exec, %[sc]\n\t" \
    "s_cbranch_execz 21f\n\t" \
    "global_store_dword %[t2" #X "], %[resq" #X "], %[mtqb]\n" \
    "21:\n\t" \
    "s_mov_b64 %[d" #X "], 0\n\t"

// ---- FETCH: start walks on the free lanes of context X with the next tile positions -----------------------------------------------
// wave-uniform: wnext/wend = the slice of tile positions this wave is handing out, exh = the tile's positions are gone.
// For a position p (tile-relative) the LDS index is pl = p + B_HIST.  Clamps (the reasons are in k_match4's fetch):
//   rem = rem0 - p = lookahead; walks need rem >= 3 (:780) and strategy != HuffmanOnly (stratm = 0 then, :786);
//   basem = LDS index of window index 1 for the position's window base (bmlo before the tile position `sw`, bmhi from it on);
//   first candidate: pl - l0 >= max(pl - MAX_DIST, basem) (:788); chain candidates: cl >= max(pl - MAX_DIST + 1, basem) (:609).
// A slice is all on one side of `sw` except one slice per 32 Ki positions: bms is picked per visit by scalar code and the mixed
// slice takes the per-lane select.  Lookahead clamps only matter in a tile that ends within 258 + slice of the segment's end.
// A walk that has found nothing yet keeps best_len = 1, not the reference's 2 (:485 Math.Max(matchLen, MIN_MATCH - 1)): its filter bytes
// are then bytes 0 and 1 of the candidate, and with the hash equal those two decide byte 2 — h = (b0 << 10 ^ b1 << 5 ^ b2) & 0x7FFF holds
// all of b2 once b0 and b1 are given — so the filter passes exactly the candidates that share the position's three bytes.  With the
// reference's bytes 1 and 2 it also passes the chain's hash collisions that differ in the three high bits of byte 0 ('T' / 't' / '4'):
// 9 % of all compares on text, and at the end of a tile the walks that are ALL such compares (a trigram new to the window whose
// collision class is common) are the ones a CU waits for: ~110 instructions per chain step instead of ~25 (profiles/r06).  COMPLETE
// asks for a length of 3 before it records anything, so the tables are what they were.
// The first candidate shares the position's hash, so it goes straight to the compare (no filter step); like every lane that
// goes there it has already moved on to the candidate after it (cb, hop, left as a QUICK step would leave them).
#define SZL9_FETCH(X) \
    "s_or_b64 %[sc], %[q" #X "], %[v" #X "]\n\t" \
    "s_or_b64 %[sc], %[sc], %[w" #X "]\n\t" \
    "s_not_b64 %[sa], %[sc]\n\t"                                                    /* free lanes */ \
    "s_cbranch_scc0 29f\n\t" \
    "s_cmp_lg_u32 %[exh], 0\n\t" \
    "s_cbranch_scc1 29f\n\t" \
    "s_cmp_lt_i32 %[wnext], %[wend]\n\t" \
    "s_cbranch_scc1 23f\n\t" \
    "s_bcnt1_i32_b64 %[f0], %[sa]\n\t"                                             /* a new slice from the tile counter: `slice` positions, */ \
    "s_cmp_ge_i32 %[wend], %[guide]\n\t"                                           /* near the tile's end as many as there are free lanes  */ \
    "s_cselect_b32 %[f0], %[f0], %[slice]\n\t"                                     /* (the wavefronts then run out of positions together)  */ \
    "s_mov_b64 exec, 1\n\t" \
    "v_mov_b32 %[vslice], %[f0]\n\t" \
    "ds_add_rtn_u32 %[t0" #X "], %[vzero], %[vslice]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_readfirstlane_b32 %[wnext], %[t0" #X "]\n\t" \
    "s_add_i32 %[wend], %[wnext], %[f0]\n\t" \
    "s_min_i32 %[wnext], %[wnext], %[tlen]\n\t" \
    "s_min_i32 %[wend], %[wend], %[tlen]\n\t" \
    "s_cmp_lt_i32 %[wnext], %[wend]\n\t" \
    "s_cbranch_scc1 23f\n\t" \
    "s_mov_b32 %[exh], 1\n\t" \
    "s_memrealtime %[texh]\n\t"                                                   /* (the tile's timeline, read by the debug build only) */ \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "s_branch 29f\n" \
    "23:\n\t" \
    "s_mov_b64 exec, %[sa]\n\t" \
    "v_mbcnt_lo_u32_b32 %[t0" #X "], exec_lo, 0\n\t" \
    "v_mbcnt_hi_u32_b32 %[t0" #X "], exec_hi, %[t0" #X "]\n\t"                  /* rank among the free lanes */ \
    "v_add_u32 %[t0" #X "], %[wnext], %[t0" #X "]\n\t"                           /* p */ \
    "v_cmpx_gt_i32 vcc, %[wend], %[t0" #X "]\n\t" \
    "s_mov_b64 %[sa], exec\n\t"                                                     /* lanes that take a position */ \
    "v_add_u32 %[pl" #X "], " SZL9_STR(SZL9_BHIST) ", %[t0" #X "]\n\t" \
    "v_mov_b32 %[res2" #X "], 0\n\t" \
    "v_mov_b32 %[resq" #X "], 0\n\t" \
    /* lookahead: only near the segment's end */ \
    "s_sub_i32 %[f2], %[rem0], %[wend]\n\t"                                        /* smallest lookahead of the slice - 1 */ \
    "s_cmp_ge_i32 %[f2], 257\n\t" \
    "s_cbranch_scc1 24f\n\t" \
    "v_sub_u32 %[t1" #X "], %[rem0], %[t0" #X "]\n\t" \
    "v_min_i32 %[cap" #X "], 0x102, %[t1" #X "]\n\t"                               /* scanMax :479 */ \
    "v_min_i32 %[nice" #X "], %[nicel], %[t1" #X "]\n\t"                           /* :485 */ \
    "v_cmpx_le_i32 vcc, 3, %[t1" #X "]\n\t"                                        /* :780 */ \
    "s_branch 25f\n" \
    "24:\n\t" \
    "v_mov_b32 %[cap" #X "], 0x102\n\t" \
    "v_mov_b32 %[nice" #X "], %[nicel]\n" \
    "25:\n\t" \
    "s_and_b64 exec, exec, %[stratm]\n\t" \
    "v_lshlrev_b32 %[t2" #X "], 1, %[pl" #X "]\n\t" \
    "v_and_b32 %[t5" #X "], -4, %[pl" #X "]\n\t" \
    "ds_read_u16 %[t2" #X "], %[t2" #X "] offset:" SZL9_STR(SZL9_LB) "\n\t"     /* hashHead (:782) as a distance */ \
    SZL9_FETCH_PB_READS(X) \
    "ds_read_b32 %[p0" #X "], %[t5" #X "] offset:" SZL9_STR(SZL9_D) "\n\t"      /* the position's first 16 bytes */ \
    "ds_read_b32 %[p1" #X "], %[t5" #X "] offset:" SZL9_STR(SZL9_D4) "\n\t" \
    "ds_read_b32 %[p2" #X "], %[t5" #X "] offset:" SZL9_STR(SZL9_D8) "\n\t" \
    "ds_read_b32 %[p3" #X "], %[t5" #X "] offset:" SZL9_STR(SZL9_D12) "\n\t" \
    "ds_read_b32 %[t6" #X "], %[t5" #X "] offset:" SZL9_STR(SZL9_D16) "\n\t" \
    /* window base of the slice */ \
    "s_mov_b32 %[f2], %[bmlo]\n\t" \
    "s_cmp_ge_i32 %[wnext], %[sw]\n\t" \
    "s_cselect_b32 %[f2], %[bmhi], %[f2]\n\t" \
    "v_mov_b32 %[t7" #X "], %[f2]\n\t" \
    "s_cmp_lt_i32 %[wnext], %[sw]\n\t" \
    "s_cselect_b32 %[f0], 1, 0\n\t" \
    "s_cmp_gt_i32 %[wend], %[sw]\n\t" \
    "s_cselect_b32 %[f1], 1, 0\n\t" \
    "s_and_b32 %[f0], %[f0], %[f1]\n\t" \
    "s_cmp_eq_u32 %[f0], 0\n\t" \
    "s_cbranch_scc1 26f\n\t" \
    "v_mov_b32 %[t1" #X "], %[bmhi]\n\t"                                           /* the slice holds a window slide */ \
    "v_cmp_le_i32 vcc, %[sw], %[t0" #X "]\n\t" \
    "v_cndmask_b32 %[t7" #X "], %[t7" #X "], %[t1" #X "], vcc\n" \
    "26:\n\t" \
    "v_add_u32 %[t1" #X "], 6, %[t0" #X "]\n\t"                                    /* pl - MAX_DIST */ \
    "v_add_u32 %[t0" #X "], 7, %[t0" #X "]\n\t"                                    /* pl - (MAX_DIST - 1) */ \
    "v_max_i32 %[mincb" #X "], %[t0" #X "], %[t7" #X "]\n\t"                     /* chain limit */ \
    "v_max_i32 %[t1" #X "], %[t1" #X "], %[t7" #X "]\n\t"                        /* first candidate's limit */ \
    "v_add_u32 %[mincb" #X "], 1, %[mincb" #X "]\n\t"                            /* (+ the filter's best_len, 1) */ \
    "s_waitcnt lgkmcnt(" SZL9_FETCH_HEAD_WAIT ")\n\t" \
    "v_sub_u32 %[cb" #X "], %[pl" #X "], %[t2" #X "]\n\t"                        /* hashHead as an index */ \
    "v_cmpx_ge_i32 vcc, %[cb" #X "], %[t1" #X "]\n\t"                            /* strstart - hashHead <= MAX_DIST (:788) */ \
    "v_lshlrev_b32 %[t0" #X "], 1, %[cb" #X "]\n\t" \
    "ds_read_u16 %[hop" #X "], %[t0" #X "] offset:" SZL9_STR(SZL9_LB) "\n\t"    /* its prev[] hop */ \
    "v_mov_b32 %[best" #X "], 1\n\t" \
    SZL9_KD_RESET(X) \
    "v_mov_b32 %[kk" #X "], -2\n\t" \
    "v_mov_b32 %[left" #X "], %[chainm2]\n\t"                                      /* max_chain - 1 may follow the first; one is taken below */ \
    "s_waitcnt lgkmcnt(1)\n\t" \
    "v_alignbyte_b32 %[p0" #X "], %[p1" #X "], %[p0" #X "], %[pl" #X "]\n\t" \
    SZL9_PB_FIRST(X) \
    "v_alignbyte_b32 %[p1" #X "], %[p2" #X "], %[p1" #X "], %[pl" #X "]\n\t" \
    "v_alignbyte_b32 %[p2" #X "], %[p3" #X "], %[p2" #X "], %[pl" #X "]\n\t" \
    "v_alignbyte_b32 %[p3" #X "], %[t6" #X "], %[p3" #X "], %[pl" #X "]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_sub_u32 %[cb" #X "], %[cb" #X "], %[hop" #X "]\n\t"                       /* on to the second candidate */ \
    "v_add_u32 %[cb" #X "], 1, %[cb" #X "]\n\t" \
    "s_or_b64 %[v" #X "], %[v" #X "], exec\n\t" \
    "s_andn2_b64 exec, %[sa], exec\n\t"                                            /* positions without a walk: an empty entry */ \
    "s_cbranch_execz 28f\n\t" \
    "v_lshlrev_b32 %[t2" #X "], 2, %[pl" #X "]\n\t" \
    "global_store_dword %[t2" #X "], %[vzero], %[mt2b]\n" \
    "28:\n\t" \
    "s_bcnt1_i32_b64 %[f0], %[sa]\n\t" \
    "s_add_i32 %[wnext], %[wnext], %[f0]\n" \
    "29:\n\t"
#if SZL9_NQ == 3
/* the filter bytes of a walk that has found nothing yet, bytes 0 and 1 of the position (above), come out of its first dword once that is
 * aligned — p[0] into bits 0-7, p[1] into bits 16-23, v_perm_b32 — instead of two more LDS reads per fetched position (round 6) */
#define SZL9_FETCH_PB_READS(X)
#define SZL9_FETCH_HEAD_WAIT "5"
#define SZL9_PB_FIRST(X) \
    "s_mov_b32 %[f1], 0x0c010c00\n\t" \
    "v_perm_b32 %[pb" #X "], %[p0" #X "], %[p0" #X "], %[f1]\n\t"
#else
#define SZL9_FETCH_PB_READS(X) \
    "ds_read_u8 %[t4" #X "], %[pl" #X "] offset:" SZL9_STR(SZL9_D1) "\n\t"
#define SZL9_FETCH_HEAD_WAIT "6"
#define SZL9_PB_FIRST(X) "v_mov_b32 %[pb" #X "], %[t4" #X "]\n\t"
#endif

#define SZL9_BUSY(X, N) \
    "s_or_b64 %[sc], %[q" #X "], %[v" #X "]\n\t" \
    "s_or_b64 %[sc], %[sc], %[w" #X "]\n\t" \
    "s_bcnt1_i32_b64 %[" N "], %[sc]\n\t"

#if SZL9_NQ == 3
#define SZL9_QW1 "3"
#else
#define SZL9_QW1 "2"
#endif

// ---- the tail program: what a wavefront runs once the tile's positions are handed out -------------------------------------------------
// Nothing is fetched any more, so the time to the tile's end is the longest walk's dependent chain of rounds — and a round of the
// two-context loop costs the same ~300 instructions whether 120 or 12 walks are left (PMC / tools/sim_match9.py: a third of a tile's
// time passes here at 4-13 % lane occupancy).  Therefore:
//   * 40: the two-context loop without fetch, bookkeeping for it or thresholds (every round: one QUICK iteration = two chain steps,
//     then whoever left is compared and completed);
//   * 50: as soon as the walks of both contexts fit ONE context (<= 64), context B's walks MOVE into the free lanes of context A —
//     rank of the walk among B's = rank of the lane among A's free ones, lane ids through 64 bytes of LDS per wavefront, the 17 state
//     registers through ds_bpermute_b32 —
//   * 60: and the rest of the tile runs a one-context loop: half the instructions per round.
// Results are retired when the walks have moved (their lanes are reused) and at the end.
#define SZL9_MOVE8(a, b, c, d, e, f, g, h) \
    "ds_bpermute_b32 %[t0B], %[t3A], %[" a "B]\n\t" \
    "ds_bpermute_b32 %[t1B], %[t3A], %[" b "B]\n\t" \
    "ds_bpermute_b32 %[t2B], %[t3A], %[" c "B]\n\t" \
    "ds_bpermute_b32 %[t3B], %[t3A], %[" d "B]\n\t" \
    "ds_bpermute_b32 %[t4B], %[t3A], %[" e "B]\n\t" \
    "ds_bpermute_b32 %[t5B], %[t3A], %[" f "B]\n\t" \
    "ds_bpermute_b32 %[t6B], %[t3A], %[" g "B]\n\t" \
    "ds_bpermute_b32 %[t7B], %[t3A], %[" h "B]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_cndmask_b32 %[" a "A], %[" a "A], %[t0B], %[sa]\n\t" \
    "v_cndmask_b32 %[" b "A], %[" b "A], %[t1B], %[sa]\n\t" \
    "v_cndmask_b32 %[" c "A], %[" c "A], %[t2B], %[sa]\n\t" \
    "v_cndmask_b32 %[" d "A], %[" d "A], %[t3B], %[sa]\n\t" \
    "v_cndmask_b32 %[" e "A], %[" e "A], %[t4B], %[sa]\n\t" \
    "v_cndmask_b32 %[" f "A], %[" f "A], %[t5B], %[sa]\n\t" \
    "v_cndmask_b32 %[" g "A], %[" g "A], %[t6B], %[sa]\n\t" \
    "v_cndmask_b32 %[" h "A], %[" h "A], %[t7B], %[sa]\n\t"

// ---- the one-context loop with RUN-AHEAD (tailp = 2) ------------------------------------------------------------------------------
// In the plain one-context loop a wavefront's few walks take turns: two chain steps for the walkers, then — most rounds — a compare and
// its completion for the one or two lanes whose candidate passed the filter, ~110 instructions a round of which the slowest walk, the
// one the whole CU waits for, gets two steps (tools/sim_match9.py: ~11000 instructions in the tile's slowest wavefront for <= 128 steps).
// Here a lane whose candidate passes the filter does NOT stop: it notes the event (cb, hop, left -> context B's registers, free since
// the merge; mask qB = "event pending") and walks on with the best_len it has — most compares do not improve on it.  Every `ktail1`
// iterations the pending events are compared in ONE pass; an event that does improve best_len takes its lane back to where the event
// was noted (the steps since are walked again under the new best_len), any other leaves the lane where it has walked to.  A lane that
// meets a second event before the first is compared waits with it in its live registers (vA), as every lane does in the plain loop.
// The filter has a THIRD byte here (pbA bits 8-15; its offset relative to cb in mincbB): the offset at which the lane's last compare
// failed.  Walks through lines that differ from the position's in one field pass the two-byte filter at every candidate and fail every
// compare at that field (logs: 4.6 of 5.7 compares per position do not improve, 70 % of them caught by this byte; text: the last walks of
// a tile); LDS has room for the read at the end of a tile, where the chain of rounds is what counts.
// Masks: qA walking, vA waiting with an event in the live registers, qB event noted, vB walk over but an event noted, wB / cB / mB scratch.
#define SZL9_RA_STEP(L) \
    "v_add_u32 %[t4A], %[cbA], %[mincbB]\n\t"              /* the candidate's byte at the offset where the lane's last compare failed */ \
    SZL9_Q_ISSUE(A) \
    "ds_read_u8 %[t3A], %[t4A] offset:" SZL9_STR(SZL9_D) "\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    SZL9_Q_COMBINE(A) \
    "v_lshl_or_b32 %[t1A], %[t3A], 8, %[t1A]\n\t" \
    "v_sub_u32 %[cbA], %[cbA], %[hopA]\n\t" \
    "v_subrev_co_u32 %[leftA], vcc, 1, %[leftA]\n\t" \
    "s_andn2_b64 exec, exec, vcc\n\t" \
    "v_cmpx_ge_i32 vcc, %[cbA], %[mincbA]\n\t" \
    "v_cmp_eq_u32 vcc, %[pbA], %[t1A]\n\t" \
    "s_cbranch_vccz " L "f\n\t" \
    "s_and_b64 %[sa], vcc, %[qB]\n\t"                      /* passed, an event already noted: the lane waits where it stands */ \
    "s_andn2_b64 %[sc], vcc, %[qB]\n\t"                    /* passed, nothing noted yet */ \
    "s_andn2_b64 exec, exec, %[sa]\n\t" \
    "s_mov_b64 %[mB], exec\n\t" \
    "s_mov_b64 exec, %[sc]\n\t" \
    "v_mov_b32 %[cbB], %[cbA]\n\t" \
    "v_mov_b32 %[hopB], %[hopA]\n\t" \
    "v_mov_b32 %[leftB], %[leftA]\n\t" \
    "s_or_b64 %[qB], %[qB], %[sc]\n\t" \
    "s_mov_b64 exec, %[mB]\n" \
    L ":\n\t"
/* (the third filter byte = the second again: offset best_len) */
#define SZL9_RA_KRESET \
    "v_mov_b32 %[mincbB], 0\n\t" \
    "v_lshrrev_b32 %[t3A], 16, %[pbA]\n\t" \
    "v_and_b32 %[pbA], 0xffff00ff, %[pbA]\n\t" \
    "v_lshl_or_b32 %[pbA], %[t3A], 8, %[pbA]\n\t"
#define SZL9_RA_SWAP \
    "v_mov_b32 %[t5A], %[cbA]\n\t" \
    "v_mov_b32 %[t6A], %[hopA]\n\t" \
    "v_mov_b32 %[t7A], %[leftA]\n\t" \
    "v_mov_b32 %[cbA], %[cbB]\n\t" \
    "v_mov_b32 %[hopA], %[hopB]\n\t" \
    "v_mov_b32 %[leftA], %[leftB]\n\t" \
    "v_mov_b32 %[cbB], %[t5A]\n\t" \
    "v_mov_b32 %[hopB], %[t6A]\n\t" \
    "v_mov_b32 %[leftB], %[t7A]\n\t"
#define SZL9_TAIL_RA \
    "; @phase census\n" \
    "79:\n\t"                                              /* entry: a compare under way starts again as an event in the live registers */ \
    "s_or_b64 %[vA], %[vA], %[wA]\n\t" \
    "s_mov_b64 %[wA], 0\n\t" \
    "s_mov_b64 exec, -1\n\t" \
    SZL9_RA_KRESET \
    "\n" \
    "80:\n\t" \
    SZL9_BUSY(A, "n0") \
    "s_or_b64 %[sc], %[qB], %[vB]\n\t" \
    "s_cbranch_scc1 801f\n\t" \
    "s_cmp_eq_u32 %[n0], 0\n\t" \
    "s_cbranch_scc1 98f\n" \
    "801:\n\t" \
    "s_cmp_eq_u64 %[qA], 0\n\t"                            /* nobody walks: compare what is noted */ \
    "s_cbranch_scc1 84f\n\t" \
    "s_mov_b64 %[mA], %[qA]\n\t" \
    "s_mov_b32 %[kt], %[ktail1]\n\t" \
    "s_mov_b64 exec, %[qA]\n" \
    "; @phase quick\n" \
    "81:\n\t" \
    SZL9_RA_STEP("811") \
    "s_cbranch_execz 85f\n\t" \
    SZL9_RA_STEP("812") \
    "s_cbranch_execz 85f\n\t" \
    "s_sub_u32 %[kt], %[kt], 1\n\t" \
    "s_cmp_lg_u32 %[kt], 0\n\t" \
    "s_cbranch_scc1 81b\n" \
    "; @phase classify\n" \
    "85:\n\t" \
    "s_andn2_b64 %[sa], %[mA], exec\n\t"                   /* left the walk in this phase */ \
    "s_mov_b64 %[qA], exec\n\t" \
    "s_mov_b64 exec, %[sa]\n\t" \
    "s_cbranch_execz 86f\n\t" \
    "v_cmp_eq_u32 vcc, %[pbA], %[t1A]\n\t"                 /* the candidate they stood on passes the filter: an event in the live registers */ \
    "s_or_b64 %[vA], %[vA], vcc\n\t" \
    "s_andn2_b64 %[sc], exec, vcc\n\t"                     /* the others' walks are over */ \
    "s_and_b64 %[cm], %[sc], %[qB]\n\t" \
    "s_or_b64 %[vB], %[vB], %[cm]\n\t"                     /* ... unless what they noted turns out to improve best_len */ \
    "s_andn2_b64 %[sc], %[sc], %[qB]\n\t" \
    "s_or_b64 %[dA], %[dA], %[sc]\n" \
    "86:\n\t" \
    "s_or_b64 %[sc], %[qB], %[vA]\n\t" \
    "s_cbranch_scc0 80b\n" \
    /* ---- 84: one compare pass over the noted events (qB) and the events in live registers (vA) */ \
    "; @phase verify1\n" \
    "84:\n\t" \
    "s_mov_b64 %[wB], %[qB]\n\t" \
    "s_or_b64 %[cm], %[qB], %[vA]\n\t" \
    "s_mov_b64 exec, %[qB]\n\t" \
    "s_cbranch_execz 841f\n\t" \
    SZL9_RA_SWAP                                             /* the noted event into the live registers, where the lane has walked to out of the way */ \
    "\n841:\n\t" \
    "s_mov_b64 %[cA], 0\n\t" \
    "s_mov_b64 %[wA], 0\n\t" \
    "s_mov_b64 exec, %[cm]\n\t" \
    "s_cbranch_execz 843f\n\t" \
    SZL9_VF_ISSUE(A) \
    "s_waitcnt lgkmcnt(0)\n\t" \
    SZL9_VF_FINISH(A) \
    "s_mov_b64 %[wA], %[mA]\n\t" \
    "s_andn2_b64 %[cA], %[cm], %[mA]\n" \
    "; @phase verify2\n" \
    "842:\n\t" \
    "s_cmp_eq_u64 %[wA], 0\n\t" \
    "s_cbranch_scc1 843f\n\t" \
    SZL9_W_STEP(A) \
    "s_branch 842b\n" \
    "; @phase complete\n" \
    "843:\n\t" \
    "s_mov_b64 exec, %[cm]\n\t" \
    "v_min_i32 %[t2A], %[offA], %[capA]\n\t" \
    "v_cmp_gt_i32 %[sc], %[t2A], %[bestA]\n\t" \
    "v_cmp_lt_i32 vcc, 2, %[t2A]\n\t" \
    "s_and_b64 %[sc], %[sc], vcc\n\t" \
    "s_mov_b64 %[cB], %[sc]\n\t"                            /* improved */ \
    "s_mov_b64 exec, %[sc]\n\t" \
    "s_cbranch_execz 844f\n\t" \
    SZL9_IMPROVE(A) \
    SZL9_RA_KRESET \
    "\n844:\n\t"                                             /* sc = improved lanes that reached niceLength (0 if nobody improved) */ \
    "s_andn2_b64 exec, %[cm], %[cB]\n\t"                   /* compared and not improved: the offset of the mismatch becomes the third filter byte — */ \
    "s_cbranch_execz 846f\n\t"                             /* the chain's next candidates tend to fail where this one did (a field that differs */ \
    "v_min_i32 %[t2A], %[offA], %[bestA]\n\t"              /* between otherwise equal lines), and any byte up to best_len is a valid filter */ \
    "v_add_u32 %[t1A], %[plA], %[t2A]\n\t" \
    "ds_read_u8 %[t1A], %[t1A] offset:" SZL9_STR(SZL9_D) "\n\t" \
    "v_sub_u32 %[mincbB], %[t2A], %[bestA]\n\t" \
    "v_and_b32 %[pbA], 0xffff00ff, %[pbA]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_lshl_or_b32 %[pbA], %[t1A], 8, %[pbA]\n" \
    "846:\n\t" \
    "s_andn2_b64 %[sa], %[cm], %[wB]\n\t" \
    "s_or_b64 %[sa], %[sa], %[cB]\n\t"                     /* lanes whose live registers are their walk from here on: improved, or no run-ahead */ \
    "s_mov_b64 exec, %[sa]\n\t" \
    "v_cmp_lt_i32 vcc, %[cbA], %[mincbA]\n\t" \
    "s_or_b64 %[sc], %[sc], vcc\n\t" \
    "v_cmp_gt_i32 vcc, 0, %[leftA]\n\t" \
    "s_or_b64 %[sc], %[sc], vcc\n\t"                       /* ... of which these end here */ \
    "s_or_b64 %[dA], %[dA], %[sc]\n\t" \
    "s_andn2_b64 %[qA], %[qA], %[sc]\n\t" \
    "s_andn2_b64 %[sa], %[sa], %[sc]\n\t" \
    "s_or_b64 %[qA], %[qA], %[sa]\n\t" \
    "s_and_b64 %[vA], %[vA], %[wB]\n\t"                    /* still waiting with a second event: noted lanes that did not improve */ \
    "s_andn2_b64 %[vA], %[vA], %[cB]\n\t" \
    "s_andn2_b64 %[sc], %[vB], %[cB]\n\t"                  /* walk over and the noted event did not improve: done */ \
    "s_or_b64 %[dA], %[dA], %[sc]\n\t" \
    "s_mov_b64 %[vB], 0\n\t" \
    "s_andn2_b64 exec, %[wB], %[cB]\n\t" \
    "s_cbranch_execz 845f\n\t" \
    SZL9_RA_SWAP                                             /* back to where the lane had walked to */ \
    "\n845:\n\t" \
    "s_mov_b64 %[qB], 0\n\t" \
    "s_mov_b64 %[wB], 0\n\t" \
    "s_branch 80b\n"

#define SZL9_MOVE_KD_0
#define SZL9_MOVE_KD_1 \
    "ds_bpermute_b32 %[t2B], %[t3A], %[kdB]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_cndmask_b32 %[kdA], %[kdA], %[t2B], %[sa]\n\t"
#define SZL9_MOVE_KD SZL9_FORM(SZL9_MOVE_KD_)
#define SZL9_TAIL \
    "; @phase census\n" \
    "40:\n\t" \
    SZL9_BUSY(A, "n0") \
    SZL9_BUSY(B, "n1") \
    "s_add_u32 %[n2], %[n0], %[n1]\n\t" \
    "s_cmp_eq_u32 %[n2], 0\n\t" \
    "s_cbranch_scc1 98f\n\t" \
    "s_cmp_eq_u32 %[n1], 0\n\t"                          /* context B is empty: the one-context loop */ \
    "s_cbranch_scc1 60f\n\t" \
    "s_cmp_le_i32 %[n2], %[mth]\n\t"                     /* both fit one context: move B's walks over */ \
    "s_cbranch_scc1 50f\n\t" \
    "s_bcnt1_i32_b64 %[n2], %[vA]\n\t" \
    "s_bcnt1_i32_b64 %[f0], %[vB]\n\t" \
    "s_add_u32 %[n2], %[n2], %[f0]\n\t" \
    "s_cmp_ge_u32 %[n2], %[vtht]\n\t" \
    "s_cbranch_scc1 44f\n\t" \
    "s_or_b64 %[sc], %[qA], %[qB]\n\t" \
    "s_cbranch_scc0 44f\n\t" \
    "s_mov_b64 %[mA], %[qA]\n\t" \
    "s_mov_b64 %[mB], %[qB]\n\t" \
    "s_mov_b32 %[kt], %[ktail]\n" \
    "; @phase quick\n" \
    "41:\n\t" \
    "s_mov_b64 exec, %[mA]\n\t" \
    SZL9_Q_ISSUE(A) \
    "s_mov_b64 exec, %[mB]\n\t" \
    SZL9_Q_ISSUE(B) \
    "s_mov_b64 exec, %[mA]\n\t" \
    "s_waitcnt lgkmcnt(" SZL9_QW1 ")\n\t" \
    SZL9_Q_FINISH(A) \
    SZL9_Q_ISSUE(A) \
    "s_mov_b64 exec, %[mB]\n\t" \
    "s_waitcnt lgkmcnt(" SZL9_QW1 ")\n\t" \
    SZL9_Q_FINISH(B) \
    SZL9_Q_ISSUE(B) \
    "s_mov_b64 exec, %[mA]\n\t" \
    "s_waitcnt lgkmcnt(" SZL9_QW1 ")\n\t" \
    SZL9_Q_FINISH_LAST(A) \
    "s_mov_b64 exec, %[mB]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    SZL9_Q_FINISH_LAST(B) \
    "s_or_b64 %[sc], %[mA], %[mB]\n\t" \
    "s_cbranch_scc0 42f\n\t"                              /* nobody walks any more */ \
    "s_sub_u32 %[kt], %[kt], 1\n\t" \
    "s_cmp_lg_u32 %[kt], 0\n\t" \
    "s_cbranch_scc1 41b\n" \
    "; @phase classify\n" \
    "42:\n\t" \
    SZL9_Q_CLASSIFY(A) \
    SZL9_Q_CLASSIFY(B) \
    "s_branch 40b\n" \
    "; @phase verify1\n" \
    "44:\n\t" \
    "s_mov_b64 %[cA], 0\n\t" \
    "s_mov_b64 %[cB], 0\n\t" \
    "s_cmp_eq_u64 %[vB], 0\n\t" \
    "s_cbranch_scc1 46f\n\t" \
    "s_cmp_eq_u64 %[vA], 0\n\t" \
    "s_cbranch_scc1 47f\n\t" \
    "s_mov_b64 exec, %[vA]\n\t" \
    SZL9_VF_ISSUE(A) \
    "s_mov_b64 exec, %[vB]\n\t" \
    SZL9_VF_ISSUE(B) \
    "s_mov_b64 exec, %[vA]\n\t" \
    "s_waitcnt lgkmcnt(5)\n\t" \
    SZL9_VF_FINISH(A) \
    "s_mov_b64 exec, %[vB]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    SZL9_VF_FINISH(B) \
    SZL9_AFTER_VF(A) \
    SZL9_AFTER_VF(B) \
    "s_branch 48f\n" \
    "46:\n\t" \
    "s_cmp_eq_u64 %[vA], 0\n\t" \
    "s_cbranch_scc1 48f\n\t" \
    "s_mov_b64 exec, %[vA]\n\t" \
    SZL9_VF_ISSUE(A) \
    "s_waitcnt lgkmcnt(0)\n\t" \
    SZL9_VF_FINISH(A) \
    SZL9_AFTER_VF(A) \
    "s_branch 48f\n" \
    "47:\n\t" \
    "s_mov_b64 exec, %[vB]\n\t" \
    SZL9_VF_ISSUE(B) \
    "s_waitcnt lgkmcnt(0)\n\t" \
    SZL9_VF_FINISH(B) \
    SZL9_AFTER_VF(B) \
    "; @phase verify2\n" \
    "48:\n\t" \
    "s_or_b64 %[sc], %[wA], %[wB]\n\t" \
    "s_cbranch_scc0 49f\n" \
    "451:\n\t" \
    SZL9_W_STEP(A) \
    SZL9_W_STEP(B) \
    "s_or_b64 %[sc], %[qA], %[qB]\n\t" \
    "s_or_b64 %[sc], %[sc], %[cA]\n\t" \
    "s_or_b64 %[sc], %[sc], %[cB]\n\t" \
    "s_cbranch_scc1 49f\n\t" \
    "s_or_b64 %[sc], %[wA], %[wB]\n\t" \
    "s_cbranch_scc1 451b\n" \
    "; @phase complete\n" \
    "49:\n\t" \
    SZL9_COMPLETE(A) \
    SZL9_COMPLETE(B) \
    "s_branch 40b\n" \
    /* ---- 50: context B's walks move into the free lanes of context A */ \
    "; @phase merge\n" \
    "50:\n\t" \
    SZL9_RETIRE(A) \
    SZL9_RETIRE(B) \
    "s_or_b64 %[cm], %[qB], %[vB]\n\t" \
    "s_or_b64 %[cm], %[cm], %[wB]\n\t"                   /* B's walks */ \
    "s_or_b64 %[sa], %[qA], %[vA]\n\t" \
    "s_or_b64 %[sa], %[sa], %[wA]\n\t" \
    "s_not_b64 %[sa], %[sa]\n\t"                         /* A's free lanes */ \
    "s_mov_b64 exec, -1\n\t" \
    "v_mbcnt_lo_u32_b32 %[t2A], -1, 0\n\t" \
    "v_mbcnt_hi_u32_b32 %[t2A], -1, %[t2A]\n\t"         /* lane id */ \
    "s_mov_b64 exec, %[cm]\n\t" \
    "v_mbcnt_lo_u32_b32 %[t0A], exec_lo, 0\n\t" \
    "v_mbcnt_hi_u32_b32 %[t0A], exec_hi, %[t0A]\n\t"    /* rank among B's walks */ \
    "v_add_u32 %[t0A], %[wscr], %[t0A]\n\t" \
    "ds_write_b8 %[t0A], %[t2A]\n\t"                     /* scratch[rank] = lane */ \
    "s_mov_b64 exec, %[sa]\n\t" \
    "v_mbcnt_lo_u32_b32 %[t1A], exec_lo, 0\n\t" \
    "v_mbcnt_hi_u32_b32 %[t1A], exec_hi, %[t1A]\n\t"    /* rank among A's free lanes */ \
    "v_cmpx_gt_u32 vcc, %[n1], %[t1A]\n\t"               /* the first n1 of them receive */ \
    "s_mov_b64 %[sa], exec\n\t" \
    "v_add_u32 %[t1A], %[wscr], %[t1A]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "ds_read_u8 %[t3A], %[t1A]\n\t"                      /* the lane to take a walk from */ \
    "s_mov_b64 exec, -1\n\t" \
    "v_cndmask_b32 %[t4A], 0, 1, %[qB]\n\t"              /* the walk's phase: 1 walking, 2 first compare, 4 comparing on */ \
    "v_cndmask_b32 %[t5A], 0, 2, %[vB]\n\t" \
    "v_cndmask_b32 %[t6A], 0, 4, %[wB]\n\t" \
    "v_or3_b32 %[t4A], %[t4A], %[t5A], %[t6A]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_lshlrev_b32 %[t3A], 2, %[t3A]\n\t" \
    SZL9_MOVE8("pl", "cb", "kk", "mincb", "left", "pb", "best", "off") \
    SZL9_MOVE8("cap", "nice", "res2", "resq", "p0", "p1", "p2", "p3") \
    "ds_bpermute_b32 %[t0B], %[t3A], %[hopB]\n\t" \
    "ds_bpermute_b32 %[t1B], %[t3A], %[t4A]\n\t" \
    SZL9_MOVE_KD \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_cndmask_b32 %[hopA], %[hopA], %[t0B], %[sa]\n\t" \
    "s_mov_b64 exec, %[sa]\n\t" \
    "v_and_b32 %[t2B], 1, %[t1B]\n\t" \
    "v_cmp_ne_u32 vcc, 0, %[t2B]\n\t" \
    "s_or_b64 %[qA], %[qA], vcc\n\t" \
    "v_and_b32 %[t2B], 2, %[t1B]\n\t" \
    "v_cmp_ne_u32 vcc, 0, %[t2B]\n\t" \
    "s_or_b64 %[vA], %[vA], vcc\n\t" \
    "v_and_b32 %[t2B], 4, %[t1B]\n\t" \
    "v_cmp_ne_u32 vcc, 0, %[t2B]\n\t" \
    "s_or_b64 %[wA], %[wA], vcc\n\t" \
    "s_mov_b64 %[qB], 0\n\t" \
    "s_mov_b64 %[vB], 0\n\t" \
    "s_mov_b64 %[wB], 0\n" \
    /* ---- 60: the one-context loop */ \
    "; @phase census\n" \
    "60:\n\t" \
    "s_cmp_eq_u32 %[tailp], 2\n\t" \
    "s_cbranch_scc1 79f\n\t" \
    SZL9_BUSY(A, "n0") \
    "s_cmp_eq_u32 %[n0], 0\n\t" \
    "s_cbranch_scc1 98f\n\t" \
    "s_bcnt1_i32_b64 %[n2], %[vA]\n\t" \
    "s_cmp_ge_u32 %[n2], %[vtht1]\n\t" \
    "s_cbranch_scc1 64f\n\t" \
    "s_cmp_eq_u64 %[qA], 0\n\t" \
    "s_cbranch_scc1 64f\n\t" \
    "s_mov_b64 %[mA], %[qA]\n\t" \
    "s_mov_b32 %[kt], %[ktail1]\n\t" \
    "s_mov_b64 exec, %[mA]\n" \
    "; @phase quick\n" \
    "61:\n\t" \
    SZL9_Q_ISSUE(A) \
    "s_waitcnt lgkmcnt(0)\n\t" \
    SZL9_Q_FINISH_(A, "") \
    "s_cbranch_execz 62f\n\t" \
    SZL9_Q_ISSUE(A) \
    "s_waitcnt lgkmcnt(0)\n\t" \
    SZL9_Q_FINISH_(A, "") \
    "s_cbranch_execz 62f\n\t" \
    "s_sub_u32 %[kt], %[kt], 1\n\t" \
    "s_cmp_lg_u32 %[kt], 0\n\t" \
    "s_cbranch_scc1 61b\n" \
    "; @phase classify\n" \
    "62:\n\t" \
    "s_mov_b64 %[mA], exec\n\t" \
    SZL9_Q_CLASSIFY(A) \
    "s_branch 60b\n" \
    "; @phase verify1\n" \
    "64:\n\t" \
    "s_mov_b64 %[cA], 0\n\t" \
    "s_cmp_eq_u64 %[vA], 0\n\t" \
    "s_cbranch_scc1 68f\n\t" \
    "s_mov_b64 exec, %[vA]\n\t" \
    SZL9_VF_ISSUE(A) \
    "s_waitcnt lgkmcnt(0)\n\t" \
    SZL9_VF_FINISH(A) \
    SZL9_AFTER_VF(A) \
    "; @phase verify2\n" \
    "68:\n\t" \
    "s_cmp_eq_u64 %[wA], 0\n\t" \
    "s_cbranch_scc1 69f\n" \
    "651:\n\t" \
    SZL9_W_STEP(A) \
    "s_or_b64 %[sc], %[qA], %[cA]\n\t" \
    "s_cbranch_scc1 69f\n\t" \
    "s_cmp_lg_u64 %[wA], 0\n\t" \
    "s_cbranch_scc1 651b\n" \
    "; @phase complete\n" \
    "69:\n\t" \
    SZL9_COMPLETE(A) \
    "s_branch 60b\n" \
    SZL9_TAIL_RA \
    "; @phase retire\n" \
    "98:\n\t" \
    SZL9_RETIRE(A) \
    SZL9_RETIRE(B)

// ---- the main loop -----------------------------------------------------------------------------------------------------------
// thresholds (SGPR inputs): bexit = 64 - (free lanes of ONE context that make the engine stop for a fetch); vth = contexts
// waiting for their first compare that start a VERIFY pass; wth / wkeep = the same for the later compare steps; qkeep = contexts
// that keep the QUICK loop going (qkeept once the tile is handed out).  n0 / n1 = busy lanes of context A / B on entry to "1:".
#define SZL9_TEXT \
    "s_mov_b64 %[sv], exec\n\t" \
    "v_mov_b32 %[vzero], 0\n\t" \
    "v_mov_b32 %[vslice], %[slice]\n\t" \
    "s_mov_b32 %[n0], 0\n\t" \
    "s_mov_b32 %[n1], 0\n" \
    "; @phase fetch\n" \
    "1:\n\t" \
    /* ---- retire + fetch: the context with more free lanes (both when the other one has enough free lanes too) */ \
    "s_cmp_lg_u32 %[exh], 0\n\t" \
    "s_cbranch_scc1 4f\n\t" \
    "s_cmp_gt_u32 %[n0], %[n1]\n\t" \
    "s_cbranch_scc1 2f\n\t" \
    SZL9_RETIRE(A) \
    SZL9_FETCH(A) \
    "s_cmp_gt_i32 %[n1], %[bexit]\n\t" \
    "s_cbranch_scc1 5f\n" \
    "2:\n\t" \
    SZL9_RETIRE(B) \
    SZL9_FETCH(B) \
    "s_cmp_le_u32 %[n0], %[n1]\n\t"                     /* A was served first */ \
    "s_cbranch_scc1 5f\n\t" \
    "s_cmp_gt_i32 %[n0], %[bexit]\n\t" \
    "s_cbranch_scc1 5f\n\t" \
    SZL9_RETIRE(A) \
    SZL9_FETCH(A) \
    "s_branch 5f\n" \
    "4:\n\t"                                              /* the tile is handed out: only retire */ \
    SZL9_RETIRE(A) \
    SZL9_RETIRE(B) \
    "s_add_u32 %[n2], %[n0], %[n1]\n\t" \
    "s_cmp_eq_u32 %[n2], 0\n\t" \
    "s_cbranch_scc1 99f\n\t" \
    "s_mov_b32 %[qkeep], %[qkeept]\n\t" \
    "s_mov_b32 %[vth], %[vtht]\n\t" \
    "s_mov_b32 %[bexit], -1\n\t" \
    "s_cmp_lg_u32 %[tailp], 0\n\t"                     /* the tail program (SZL9_TAIL below); 0: stay in this loop (laboratory) */ \
    "s_cbranch_scc1 40f\n" \
    "5:\n" \
    "; @phase census\n" \
    "10:\n\t" \
    SZL9_BUSY(A, "n0") \
    SZL9_BUSY(B, "n1") \
    "s_add_u32 %[n2], %[n0], %[n1]\n\t" \
    "s_cmp_eq_u32 %[n2], 0\n\t"                         /* nothing in flight: fetch, or the end */ \
    "s_cbranch_scc1 1b\n\t" \
    "s_min_u32 %[n2], %[n0], %[n1]\n\t" \
    "s_cmp_le_i32 %[n2], %[bexit]\n\t"                  /* a context has enough free lanes: fetch (never once the tile is handed out) */ \
    "s_cbranch_scc1 1b\n\t" \
    "s_bcnt1_i32_b64 %[n2], %[vA]\n\t" \
    "s_bcnt1_i32_b64 %[f0], %[vB]\n\t" \
    "s_add_u32 %[n2], %[n2], %[f0]\n\t"                 /* contexts waiting for their first compare */ \
    "s_cmp_ge_u32 %[n2], %[vth]\n\t" \
    "s_cbranch_scc1 14f\n\t" \
    "s_or_b64 %[sc], %[qA], %[qB]\n\t" \
    "s_cbranch_scc0 14f\n\t"                             /* nothing in QUICK */ \
    /* ---- QUICK phase */ \
    "s_mov_b64 %[mA], %[qA]\n\t" \
    "s_mov_b64 %[mB], %[qB]\n\t" \
    "s_mov_b32 %[kt], %[ktail]\n" \
    "; @phase quick\n" \
    "11:\n\t" \
    "s_mov_b64 exec, %[mA]\n\t" \
    SZL9_Q_ISSUE(A) \
    "s_mov_b64 exec, %[mB]\n\t" \
    SZL9_Q_ISSUE(B) \
    "s_mov_b64 exec, %[mA]\n\t" \
    "s_waitcnt lgkmcnt(" SZL9_QW1 ")\n\t" \
    SZL9_Q_FINISH(A) \
    SZL9_Q_ISSUE(A)                                       /* A's next step is in flight while B finishes */ \
    "s_mov_b64 exec, %[mB]\n\t" \
    "s_waitcnt lgkmcnt(" SZL9_QW1 ")\n\t" \
    SZL9_Q_FINISH(B) \
    SZL9_Q_ISSUE(B) \
    "s_mov_b64 exec, %[mA]\n\t" \
    "s_waitcnt lgkmcnt(" SZL9_QW1 ")\n\t" \
    SZL9_Q_FINISH_LAST(A) \
    "s_mov_b64 exec, %[mB]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    SZL9_Q_FINISH_LAST(B) \
    "s_bcnt1_i32_b64 %[n0], %[mA]\n\t" \
    "s_bcnt1_i32_b64 %[n1], %[mB]\n\t" \
    "s_add_u32 %[n0], %[n0], %[n1]\n\t" \
    "s_cmp_ge_u32 %[n0], %[qkeep]\n\t" \
    "s_cbranch_scc1 11b\n\t" \
    /* once the tile is handed out nothing waits to be fetched and the time to the tile's end is the longest walk's: its lanes */ \
    /* walk `ktail` iterations between two looks at the lanes that left (every look costs the walkers ~100 instructions) */ \
    "s_cmp_lg_u32 %[exh], 0\n\t" \
    "s_cbranch_scc0 12f\n\t" \
    "s_cmp_eq_u32 %[n0], 0\n\t" \
    "s_cbranch_scc1 12f\n\t" \
    "s_sub_u32 %[kt], %[kt], 1\n\t" \
    "s_cmp_lg_u32 %[kt], 0\n\t" \
    "s_cbranch_scc1 11b\n" \
    "; @phase classify\n" \
    "12:\n\t" \
    SZL9_Q_CLASSIFY(A) \
    SZL9_Q_CLASSIFY(B) \
    "s_branch 10b\n" \
    /* ---- VERIFY: the first (16-byte) step of the new arrivals, one later step for whoever is still comparing, and the */ \
    /* completion of every compare that ended */ \
    "; @phase verify1\n" \
    "14:\n\t" \
    "s_mov_b64 %[cA], 0\n\t" \
    "s_mov_b64 %[cB], 0\n\t" \
    "s_cmp_eq_u64 %[vB], 0\n\t" \
    "s_cbranch_scc1 16f\n\t" \
    "s_cmp_eq_u64 %[vA], 0\n\t" \
    "s_cbranch_scc1 17f\n\t" \
    "s_mov_b64 exec, %[vA]\n\t" \
    SZL9_VF_ISSUE(A) \
    "s_mov_b64 exec, %[vB]\n\t" \
    SZL9_VF_ISSUE(B) \
    "s_mov_b64 exec, %[vA]\n\t" \
    "s_waitcnt lgkmcnt(5)\n\t" \
    SZL9_VF_FINISH(A) \
    "s_mov_b64 exec, %[vB]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    SZL9_VF_FINISH(B) \
    SZL9_AFTER_VF(A) \
    SZL9_AFTER_VF(B) \
    "s_branch 18f\n" \
    "16:\n\t"                                             /* only context A has new candidates */ \
    "s_cmp_eq_u64 %[vA], 0\n\t" \
    "s_cbranch_scc1 18f\n\t" \
    "s_mov_b64 exec, %[vA]\n\t" \
    SZL9_VF_ISSUE(A) \
    "s_waitcnt lgkmcnt(0)\n\t" \
    SZL9_VF_FINISH(A) \
    SZL9_AFTER_VF(A) \
    "s_branch 18f\n" \
    "17:\n\t"                                             /* only context B */ \
    "s_mov_b64 exec, %[vB]\n\t" \
    SZL9_VF_ISSUE(B) \
    "s_waitcnt lgkmcnt(0)\n\t" \
    SZL9_VF_FINISH(B) \
    SZL9_AFTER_VF(B) \
    "; @phase verify2\n" \
    "18:\n\t" \
    "s_or_b64 %[sc], %[wA], %[wB]\n\t" \
    "s_cbranch_scc0 19f\n" \
    "151:\n\t" \
    SZL9_W_STEP(A) \
    SZL9_W_STEP(B) \
    /* again right away only when every lane in flight is waiting on a long compare */ \
    "s_or_b64 %[sc], %[qA], %[qB]\n\t" \
    "s_or_b64 %[sc], %[sc], %[cA]\n\t" \
    "s_or_b64 %[sc], %[sc], %[cB]\n\t" \
    "s_cbranch_scc1 19f\n\t" \
    "s_or_b64 %[sc], %[wA], %[wB]\n\t" \
    "s_cbranch_scc1 151b\n" \
    "; @phase complete\n" \
    "19:\n\t" \
    SZL9_COMPLETE(A) \
    SZL9_COMPLETE(B) \
    "s_branch 10b\n" \
    SZL9_TAIL \
    "99:\n\t" \
    "s_mov_b64 exec, %[sv]\n\t"
