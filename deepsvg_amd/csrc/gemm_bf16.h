// Shared declarations of the bf16 GEMM kernels (gemm_bf16.hip: register-staged, any shape; gemm_bf16_glds.hip:
// LDS-DMA staged 128x128 tiles, aligned shapes).
#pragma once
#include "dsvg_common.h"
#include "../../include/dsvg.h"

enum {
    EPI_GENERIC = 0,        // everything decided at run time (any alignment, any option)
    EPI_BIAS = 1,           // C = acc [+ bias]
    EPI_BIAS_RES_DROP = 2,  // C = res + drop(acc [+ bias])
    EPI_BIAS_RELU_DROP = 3, // C = drop(relu(acc [+ bias]))
    EPI_GATE = 4,           // C = gate > 0 ? acc * gate_scale : 0
    EPI_PARTIAL = 5,        // split-K slice: raw fp32 accumulators to the workspace, nothing else
};

// Launches the LDS-DMA kernel when the call is eligible (see gemm_bf16_glds.hip); returns false otherwise.
// part_is_bf16 (split-K only, may be NULL): set to 1 when the slices were written as packed bf16 (see the kernel)
bool dsvg_gemm_bf16_glds_try(const dsvg_gemm_desc& d, int epi, dim3 grid, int tiles_n, int nwg, int k_chunk,
                             float* part, float* rs_part, int mode, hipStream_t st, int* part_is_bf16);
// launch the weight-gradient GEMMs queued on this stream under dsvg_gemm_group_scope (gemm_bf16_glds.hip); 0 if none
int dsvg_gemm_group_flush(hipStream_t st);
