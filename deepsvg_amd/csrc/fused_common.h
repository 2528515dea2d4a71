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

// LayerNorm statistics of a row whose 256 bf16 values sit packed in the registers of a lane pair (16 x uint4 per lane): sum
// and sum of squares straight from the packed words - v_dot2c_f32_bf16 with (1, 1) and with the word itself, one instruction
// per element, no unpack.  The prologues / epilogues of the token-stationary kernels are instruction-issue-bound (a wave issues
// one VALU instruction every ~5 cycles with at most two waves per SIMD and no MFMA beside them, scripts/probes/
// valu_rate_probe.hip): the two-pass form cost 5 instructions per element.  var = E[x^2] - mean^2 in fp32: the inputs are
// bf16, whose own rounding (2^-9 relative) is coarser than the cancellation error of that form for any |mean| / sigma the
// format can represent.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ln_stats_packed(const uint4 (&raw)[16], float& s, float& q) {
    const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
    s = 0.f;
    q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t w[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bf16x2_t a = __builtin_bit_cast(bf16x2_t, w[e]);
            s = __builtin_amdgcn_fdot2_f32_bf16(a, ones, s, false);
            q = __builtin_amdgcn_fdot2_f32_bf16(a, a, q, false);
        }
    }
}
// mean and rstd of a 256-column row from the two lane halves' partial sums
__device__ __forceinline__ void ln_mean_rstd256(float s, float q, float eps, float& mean, float& rstd) {
    s += __shfl_xor(s, 32, 64);
    q += __shfl_xor(q, 32, 64);
    mean = s * (1.f / 256.f);
    rstd = rsqrtf(fmaxf(q * (1.f / 256.f) - mean * mean, 0.f) + eps);
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

// the same with the global address as (wave-uniform 64-bit base in SGPRs) + (32-bit per-lane byte offset): no 64-bit VGPR
// address pair to keep alive across a loop, the per-chunk advance is scalar arithmetic
__device__ __forceinline__ void dma4s(const void* base_uniform, uint32_t lane_off, uint32_t lds) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
        "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep) : "v"(lane_off), "s"(base_uniform), "s"(lds) : "memory");
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
