// Library primitives: radix sort (Morton keys; pair keys on the rare fallback path) and exclusive scan.
// Plain library primitive (rocPRIM device radix sort) — not a hot-path kernel.
#include "ctx.hpp"
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace eh {

size_t sort_temp_bytes(uint32_t max_items) {
    size_t a = 0, b = 0;
    (void)rocprim::radix_sort_keys<rocprim::default_config, const uint64_t *, uint64_t *>(nullptr, a, nullptr, nullptr, max_items, 0, 64, nullptr);
    size_t d = 0;
    (void)rocprim::exclusive_scan(nullptr, d, (const uint32_t *)nullptr, (uint32_t *)nullptr, 0u, (size_t)max_items, rocprim::plus<uint32_t>(), nullptr);
    size_t m = a > b ? a : b;
    return (m > d ? m : d) + 256;
}

int scan_u32(edynhip_ctx *c, const uint32_t *in, uint32_t *out, uint32_t n) {
    if (n == 0) return EDYNHIP_OK;
    size_t bytes = c->sort_tmp_bytes;
    EH_HIP(c, rocprim::exclusive_scan(c->sort_tmp, bytes, in, out, 0u, (size_t)n, rocprim::plus<uint32_t>(), c->stream));
    return EDYNHIP_OK;
}

int sort_u64(edynhip_ctx *c, const uint64_t *in, uint64_t *out, uint32_t n, int begin_bit, int end_bit) {
    if (n == 0) return EDYNHIP_OK;
    size_t bytes = c->sort_tmp_bytes;
    EH_HIP(c, rocprim::radix_sort_keys(c->sort_tmp, bytes, in, out, (size_t)n, (unsigned)begin_bit, (unsigned)end_bit, c->stream));
    return EDYNHIP_OK;
}

}  // namespace eh
