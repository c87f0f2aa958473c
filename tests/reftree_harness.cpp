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
