// szl_api_inflate.hip — Inflater entry points of include/szl.h.
#include <hip/hip_runtime.h>
#include "szl_engine.h"
using namespace szl;
struct szl_inflater { int no_header; };
extern "C" {
szl_inflater *szl_inflater_create(int) { set_error("device inflate not built yet"); return nullptr; }
void szl_inflater_destroy(szl_inflater *) {}
int szl_inflater_reset(szl_inflater *) { return SZL_E_UNSUPPORTED; }
int szl_inflater_set_input(szl_inflater *, const uint8_t *, int) { return SZL_E_UNSUPPORTED; }
int szl_inflater_set_dictionary(szl_inflater *, const uint8_t *, int) { return SZL_E_UNSUPPORTED; }
int szl_inflater_inflate(szl_inflater *, uint8_t *, int) { return SZL_E_UNSUPPORTED; }
int szl_inflater_needs_input(const szl_inflater *) { return 0; }
int szl_inflater_needs_dictionary(const szl_inflater *) { return 0; }
int szl_inflater_is_finished(const szl_inflater *) { return 0; }
int szl_inflater_remaining_input(const szl_inflater *) { return 0; }
int64_t szl_inflater_total_in(const szl_inflater *) { return 0; }
int64_t szl_inflater_total_out(const szl_inflater *) { return 0; }
uint32_t szl_inflater_adler(const szl_inflater *) { return 0; }
int szl_inflate_batch_device(szl_engine *, const void *, void *, szl_stream *, size_t, unsigned, void *) { return SZL_E_UNSUPPORTED; }
int szl_inflate_batch_host(szl_engine *, const void *, void *, szl_stream *, size_t, unsigned) { return SZL_E_UNSUPPORTED; }
}
