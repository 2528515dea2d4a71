// Helpers shared by the fused token-stationary kernels (ffn_fused.hip, attn_fused.hip): packed MFMA fragment images
// streamed through an LDS ring by LDS-DMA, bf16 <-> fp32 register packing, accumulator-tile transposition.
#pragma once
#include "dsvg_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int FRAG = 1024;              // bytes per packed MFMA fragment (64 lanes x 16 B)

#define DSVG_LDS_PTR(p) ((void __attribute__((address_space(3)))*)(p))

union Frag8 {
    bf16x8 v;
    uint4 u;
};

__device__ __forceinline__ void unpack8(const uint4& t, float (&v)[8]) {
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[2 * e] = __uint_as_float(w[e] << 16);
        v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
    }
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
    return make_uint4(f2bf_pk(v[0], v[1]), f2bf_pk(v[2], v[3]), f2bf_pk(v[4], v[5]), f2bf_pk(v[6], v[7]));
}



// ---------------------------------------------------------------------------------------------------------------------
// LDS-DMA of one wave's 4 consecutive 1 KiB pieces (hidden from hipcc, see the header).  src: per-lane address of the
// first piece (+ lane * 16 included); lds: wave-uniform LDS byte address of the first piece.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dma2(const void* src, uint32_t lds) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "global_load_lds_dwordx4 %1, off offset:1024\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep) : "v"(src), "s"(lds) : "memory");
}
__device__ __forceinline__ void dma4(const void* src, uint32_t lds) {
    // the instruction's immediate offset advances BOTH addresses (global: src + offset, LDS: M0 + offset + lane * 16)
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "global_load_lds_dwordx4 %1, off offset:1024\n\t"
        "global_load_lds_dwordx4 %1, off offset:2048\n\t"
        "global_load_lds_dwordx4 %1, off offset:3072\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep) : "v"(src), "s"(lds) : "memory");
}

// One transposed 32 x 32 accumulator tile (lane: token row lane & 31, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the
// tile) -> the lane's 16 consecutive tile columns 16 (lane >> 5) .. + 15 as x[q][e] = column 4 q + e.
__device__ __forceinline__ void tile_to_cols16(const floatx16& c, uint32_t (&x)[4][4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) x[q][e] = __float_as_uint(c[4 * q + e]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        auto s01 = __builtin_amdgcn_permlane32_swap(x[0][e], x[1][e], false, false);
        auto s23 = __builtin_amdgcn_permlane32_swap(x[2][e], x[3][e], false, false);
        x[0][e] = s01[0]; x[1][e] = s01[1]; x[2][e] = s23[0]; x[3][e] = s23[1];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        auto s02 = __builtin_amdgcn_permlane32_swap(x[0][e], x[2][e], false, false);
        auto s13 = __builtin_amdgcn_permlane32_swap(x[1][e], x[3][e], false, false);
        x[0][e] = s02[0]; x[2][e] = s02[1]; x[1][e] = s13[0]; x[3][e] = s13[1];
    }
}

}  // namespace
