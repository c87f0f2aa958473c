// CPU build of csrc/szl_inflate_reftree.h for tests/test_reftree.py (the header is plain C++; k_inflate compiles the same text).
#include <stdint.h>
#include "../sharpziplib_amd/csrc/szl_inflate_reftree.h"
extern "C" int rt_table(const uint8_t *lens, int n, int16_t *out, int cap) {
    uint32_t blc[16], nxt[16];
    return szl::rt_build(lens, n, out, cap, blc, nxt);
}
extern "C" int rt_symbol(const int16_t *tree, uint32_t bits, uint32_t avail) { return szl::rt_get_symbol(tree, bits, avail); }
extern "C" int rt_quirk(const uint8_t *lens, int n) {
    uint32_t cnt[16] = {0};
    for (int i = 0; i < n; i++) cnt[lens[i]]++;
    cnt[0] = 0;
    return szl::rt_is_quirk_set(cnt) ? 1 : 0;
}

// the same script of bit-buffer operations as oracle/szl_inflate_oracle.c szo_sm_script, on the device's emulation (ExSM)
extern "C" int rt_sm_script(const uint8_t *buf, int n, const int32_t *ops, int nops, const int16_t *tree, int32_t *results) {
    szl::ExSM sm;
    sm.in = buf; sm.we = (uint64_t)n;
    const uint32_t r = (8u * (uint32_t)(n & 1)) & 15u;      // k_inflate_exact's start at bit 0: the odd first byte of SetInput is in the buffer
    sm.ws = r >> 3; sm.bits = (int32_t)r; sm.lazy = 0; sm.dirty = 0;
    sm.buffer = r ? buf[0] : 0u;
    for (int i = 0; i < nops; i++) {
        const int op = ops[2 * i], arg = ops[2 * i + 1];
        int v = 0;
        switch (op) {
        case 0: v = szl::ex_peek(sm, arg); break;
        case 1: szl::ex_drop(sm, arg); break;
        case 2: sm.buffer >>= (sm.bits & 7); sm.bits &= ~7; break;
        case 3: v = sm.bits; break;
        case 4: v = (int)szl::ex_available_bytes(sm); break;
        case 5: { v = szl::ex_get_symbol(tree, sm); if (v == -2) v = -100 + (-8); } break;   // (the oracle reports "invalid codelength 0" as -100 + SZO_ERR_CODELEN_ZERO)
        default: return -2;
        }
        results[i] = v;
    }
    return 0;
}
