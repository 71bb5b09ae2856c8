// wire.cpp — packed wire headers (include/bsx.h: bsx_pack_headers / bsx_unpack_headers).  Host code, no HIP.
//
// A Tendermint header is 14 protobuf-encoded fields, ~394 bytes together (what get_signed_header_range,
// circuits/input.rs:120-145, fetches per block).  The device works on 512-byte bsx_header records (every field padded to a fixed
// 4-byte aligned capacity); a host that ships one range per call pays PCIe for the padding — 23 % of every upload.  The packed block
// carries the same 14 fields back to back:
//
//   u32 n_headers, u32 n_bytes (whole block)            8 bytes
//   u32 off[n_headers + 1]                              byte offset of header i inside the data section; off[n] = data bytes
//   (padding to a multiple of 16)
//   data: per header  u8 len[14], then field 0 .. field 13, len[f] bytes each
//
// Nothing is interpreted here: a length over its field's capacity travels as it is and is flagged where the 512-byte record's would
// be (BSX_ERR_BAD_HEADER from the hashing kernel); only the block's own arithmetic (offsets monotone, inside the block, >= 14 apart)
// is checked on the host, because the device indexes with it.
#include <cstdint>
#include <cstring>

#include "../../include/bsx.h"

namespace {
constexpr uint32_t kAt[BSX_HEADER_FIELDS + 1] = {16, 40, 92, 104, 124, 200, 236, 272, 308, 344, 380, 416, 452, 488, 512};
inline uint64_t data_at(uint64_t n) { return (8 + 4 * (n + 1) + 15) & ~(uint64_t)15; }
}  // namespace

extern "C" {

uint64_t bsx_packed_headers_bound(uint64_t n_headers) { return data_at(n_headers) + n_headers * (BSX_HEADER_FIELDS + 512 - 16); }

int bsx_pack_headers(const bsx_header* headers, uint64_t n_headers, void* out, uint64_t out_cap, uint64_t* out_bytes) {
    if (!headers || !out || !out_bytes || n_headers == 0 || n_headers > 0x00ffffffu) return BSX_ERR_BAD_ARG;
    const uint64_t d0 = data_at(n_headers);
    if (out_cap < d0) return BSX_ERR_BAD_ARG;
    uint8_t* o = static_cast<uint8_t*>(out);
    uint32_t* off = reinterpret_cast<uint32_t*>(o + 8);
    uint64_t at = 0;
    for (uint64_t i = 0; i < n_headers; i++) {
        const uint8_t* h = reinterpret_cast<const uint8_t*>(headers + i);
        uint64_t need = BSX_HEADER_FIELDS;
        for (int f = 0; f < BSX_HEADER_FIELDS; f++) {
            const uint32_t cap = kAt[f + 1] - kAt[f];
            need += h[f] < cap ? h[f] : cap;                 // an over-long field: its length byte travels, its bytes stop at the capacity
        }
        if (d0 + at + need > out_cap) return BSX_ERR_BAD_ARG;
        off[i] = (uint32_t)at;
        uint8_t* p = o + d0 + at;
        memcpy(p, h, BSX_HEADER_FIELDS);
        p += BSX_HEADER_FIELDS;
        for (int f = 0; f < BSX_HEADER_FIELDS; f++) {
            const uint32_t cap = kAt[f + 1] - kAt[f], l = h[f] < cap ? h[f] : cap;
            memcpy(p, h + kAt[f], l);
            p += l;
        }
        at += need;
    }
    off[n_headers] = (uint32_t)at;
    memset(o + 8 + 4 * (n_headers + 1), 0, d0 - (8 + 4 * (n_headers + 1)));
    const uint32_t n32 = (uint32_t)n_headers, b32 = (uint32_t)(d0 + at);
    memcpy(o, &n32, 4);
    memcpy(o + 4, &b32, 4);
    *out_bytes = d0 + at;
    return BSX_OK;
}

// block arithmetic only; *out_n = headers in the block
int bsx_packed_headers_check(const void* packed, uint64_t packed_bytes, uint64_t* out_n) {
    if (!packed || packed_bytes < 16) return BSX_ERR_BAD_ARG;
    const uint8_t* o = static_cast<const uint8_t*>(packed);
    uint32_t n, nb;
    memcpy(&n, o, 4);
    memcpy(&nb, o + 4, 4);
    if (n == 0 || n > 0x00ffffffu || nb > packed_bytes) return BSX_ERR_BAD_HEADER;
    const uint64_t d0 = data_at(n);
    if (d0 > nb) return BSX_ERR_BAD_HEADER;
    const uint64_t data = nb - d0;
    uint32_t prev;
    memcpy(&prev, o + 8, 4);
    if (prev != 0) return BSX_ERR_BAD_HEADER;
    for (uint64_t i = 1; i <= n; i++) {
        uint32_t x;
        memcpy(&x, o + 8 + 4 * i, 4);
        if (x < prev + BSX_HEADER_FIELDS || x > data) return BSX_ERR_BAD_HEADER;
        prev = x;
    }
    if (out_n) *out_n = n;
    return BSX_OK;
}

int bsx_unpack_headers(const void* packed, uint64_t packed_bytes, bsx_header* out, uint64_t out_cap_headers, uint64_t* out_n) {
    uint64_t n = 0;
    const int rc = bsx_packed_headers_check(packed, packed_bytes, &n);
    if (rc != BSX_OK) return rc;
    if (!out && out_cap_headers == 0) {              // validation only
        if (out_n) *out_n = n;
        return BSX_OK;
    }
    if (!out || out_cap_headers < n) return BSX_ERR_BAD_ARG;
    const uint8_t* o = static_cast<const uint8_t*>(packed);
    const uint64_t d0 = data_at(n);
    for (uint64_t i = 0; i < n; i++) {
        uint32_t b0, b1;
        memcpy(&b0, o + 8 + 4 * i, 4);
        memcpy(&b1, o + 8 + 4 * (i + 1), 4);
        const uint8_t* src = o + d0 + b0;
        const uint32_t avail = b1 - b0;
        uint8_t* h = reinterpret_cast<uint8_t*>(out + i);
        memset(h, 0, sizeof(bsx_header));
        memcpy(h, src, BSX_HEADER_FIELDS);
        uint32_t at = BSX_HEADER_FIELDS;
        for (int f = 0; f < BSX_HEADER_FIELDS; f++) {
            const uint32_t cap = kAt[f + 1] - kAt[f];
            for (uint32_t k = 0; k < src[f] && k < cap; k++)
                if (at + k < avail) h[kAt[f] + k] = src[at + k];
            at += src[f];
        }
    }
    if (out_n) *out_n = n;
    return BSX_OK;
}

}  // extern "C"
