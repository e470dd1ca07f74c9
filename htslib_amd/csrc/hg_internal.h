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
    void *d_scratch[4];
    size_t d_scratch_cap[4];
};

namespace hg {
int launch_bgzf_inflate(hg_ctx *ctx, const void *d_comp, size_t comp_len, const hg_bgzf_desc *d_desc,
                        size_t nblocks, void *d_out, size_t out_cap, int32_t *d_status, hipStream_t s);
int launch_crc32(hg_ctx *ctx, const void *d_data, const uint64_t *d_off, const uint32_t *d_len, size_t n,
                 uint32_t *d_crc, hipStream_t s);
}
