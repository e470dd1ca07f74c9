// hg_internal.h -- host-side internals shared by the C-ABI shim and the kernel launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "htsgpu.h"

struct hg_ctx {
    int device;
    int cus;
    int waves_per_launch;     // resident wavefronts the persistent kernels are sized for
    unsigned int *d_ticket;   // work-queue counter (device)
    // scratch for the host-buffer convenience entry points (grown on demand)
    void *d_scratch[8];
    size_t d_scratch_cap[8];
    void *d_tok;              // deflate token lists, 256 KiB per resident workgroup
    size_t d_tok_cap;
};

namespace hg {
int launch_bgzf_inflate(hg_ctx *ctx, const void *d_comp, size_t comp_len, const hg_bgzf_desc *d_desc,
                        size_t nblocks, void *d_out, size_t out_cap, int32_t *d_status, hipStream_t s, int mode = 0);
int launch_bgzf_deflate(hg_ctx *ctx, const void *d_plain, const hg_bgzf_desc *d_desc, size_t nblocks, int level,
                        void *d_slots, uint32_t *d_clen, hipStream_t s, int mode = 0, uint32_t *d_crc = nullptr);
int launch_bgzf_pack(hg_ctx *ctx, const void *d_slots, const hg_bgzf_desc *d_desc, const uint32_t *d_clen,
                     size_t nblocks, void *d_packed, size_t cap, uint64_t *d_poff, uint64_t *d_total, int add_eof,
                     hipStream_t s);
int launch_rans4x8_decode(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, size_t n, void *d_out,
                          int32_t *d_status, uint32_t *d_scratch, hipStream_t s);
int launch_ransnx16_decode(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint32_t *d_sel4,
                           size_t n4, const uint32_t *d_sel32, size_t n32, void *d_out, int32_t *d_status,
                           uint32_t *d_scratch, hipStream_t s);
uint32_t ransnx16_enc_scratch_words(uint32_t flags);
int launch_ransnx16_encode(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint8_t *d_flags,
                           const uint32_t *d_sel4, size_t n4, const uint32_t *d_sel32, size_t n32, void *d_out,
                           uint32_t *d_out_len, void *d_wbuf, uint32_t *d_scratch, hipStream_t s);
uint32_t rans4x8_enc_scratch_words(uint32_t order);
int launch_rans4x8_encode(hg_ctx *ctx, const void *d_in, const hg_stream_desc *d_desc, const uint8_t *d_order, size_t n,
                          void *d_out, uint32_t *d_out_len, void *d_wbuf, uint32_t *d_scratch, hipStream_t s);
int launch_crc32(hg_ctx *ctx, const void *d_data, const uint64_t *d_off, const uint32_t *d_len, size_t n,
                 uint32_t *d_crc, hipStream_t s);
}
