// One launch per transformer layer for the SHORT-sequence stages of the hierarchical model - the two "group" stages
// (hierarchical_encoder, deepsvg/model/model.py:153-161; hierarchical_decoder, :246-254): N x 8 rows of d_model 256,
// sequences of 8 group tokens - forward and backward of the whole pre-LN block
//     x1 = x  + drop( Wo . MHA( LN1(x) ) + bo ) [+ drop( g[sequence] )]          improved_transformer.py:43-49,127-136
//     x2 = x1 + drop( W2 . drop( relu( W1 . LN2(x1) + b1 ) ) + b2 )                                      :51-53,138-140
// bf16 storage, fp32 accumulation and statistics.  These stages are 3 % of the model's FLOPs but ran as ~33 launches of
// 5-12 us per layer and direction (4096 rows cannot fill 256 CUs with token-stationary 256-row workgroups, and every
// unfused GEMM of this size is one exposed memory round trip per K step).
//
// Decomposition (both kernels): a 512-thread workgroup owns ONE tile of 32 rows (32 / S whole sequences), its 8 waves split
// the OUTPUT FEATURES of every matrix product - wave w computes the 32-wide column block w (its attention head, in the
// in_proj / attention phases) - so a wave issues ~130 MFMAs per layer and the launch is bound by what every CU has to
// ingest: the layer's 1 MiB of weights.  They stream from L2 straight into MFMA A-operand registers through a per-wave
// prefetch ring (dsvg_gs_pack lays them out wave-major in consumption order: one contiguous 128 KiB stream per wave
// and layer, independent of the activations, so it runs ahead across every phase boundary).  Activations move between
// the phases through row-major LDS images [32 rows][features] whose row stride is an odd number of 16-byte units:
// every MFMA B operand is one conflict-free ds_read_b128 (`row_frag`), every result tile goes back as four 8-byte
// pieces per lane (`stage_rows`), V^T / dO^T / column sums over tokens use hardware-transposed reads (`col_frag`), and
// everything the other direction or the weight-gradient GEMMs need leaves the chip as full rows, cooperatively, 16 bytes
// per lane (`store_image`).
// The attention phases are the bodies of attention_mfma.hip's MODE 2 kernels (same dropout draws, same masks).
// Dropout draws are the library's standard ones at every site, so the fused and the unfused launches are
// interchangeable between forward and backward (tests/test_kernels_gpu.py runs all four combinations).
#include "fused_common.h"
#include "pack_images.h"
#include "../../include/dsvg.h"

typedef short shortx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int GD = 256;                 // d_model
constexpr int GF = 512;                 // dim_feedforward
constexpr int GH = 8;                   // heads = waves per workgroup
constexpr int LDX = GD + 8;             // row stride (elements) of a [32][256] image: 528 B = 33 x 16 B
constexpr int LDQ = 3 * GD + 8;         // q|k|v image: 1552 B = 97 x 16 B
constexpr int LDH = GF + 8;             // hidden image: 1040 B = 65 x 16 B
constexpr int GS_FRAGS = 128;           // weight fragments per wave and layer (both directions)
constexpr int GS_PF_DEFAULT = 12;       // prefetch distance of the weight stream (fragments = KiB in flight per wave)

using dsvg_pack::rowmap;
static_assert(dsvg_pack::GS_FRAGS == GS_FRAGS && dsvg_pack::D == GD && dsvg_pack::F == GF && dsvg_pack::H == GH, "pack_images.h restates these");

// weight packing (dsvg_gs_pack): fp32 master parameters -> bf16 MFMA A fragments, wave-major, in consumption order; body
// and layout in pack_images.h
__global__ __launch_bounds__(256) void gs_pack_kernel(const float* __restrict__ flat, const int64_t* __restrict__ offs,
                                                      int n_layers, bf16_t* __restrict__ fwd, bf16_t* __restrict__ bwd) {
    dsvg_pack::gs_slot((long long)blockIdx.x * 256 + threadIdx.x, flat, offs, n_layers, fwd, bwd);
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS image helpers
// ---------------------------------------------------------------------------------------------------------------------
// B operand: B[k = 16 step + 8 h2 + e][column = row `row` of the image] - one 16-byte read
__device__ __forceinline__ bf16x8 row_frag(const bf16_t* img, int ld, int row, int col0, int step, int h2) {
    Frag8 f;
    f.u = *reinterpret_cast<const uint4*>(&img[row * ld + col0 + 16 * step + 8 * h2]);
    return f.v;
}
// A operand: A[i = column col0 + (lane & 31)][k slot e] = img[row rowmap(8 ks + e, lane >> 5)][that column]
__device__ __forceinline__ bf16x8 col_frag(const bf16_t* img, int ld, int col0, int ks, int lane) {
    const int g = lane >> 4, q16 = lane & 15;
    const int row = 16 * ks + 4 * (g >> 1) + (q16 >> 2);
    const int col = col0 + 16 * (g & 1) + 4 * (q16 & 3);
    union { bf16x8 v; shortx4 h[2]; } f;
    f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((shortx4 __attribute__((address_space(3)))*)(&img[row * ld + col]));
    f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((shortx4 __attribute__((address_space(3)))*)(&img[(row + 8) * ld + col]));
    return f.v;
}
__device__ __forceinline__ bf16x8 pack_regs(const float (&p)[16], int ks) {
    Frag8 f;
    f.u = make_uint4(f2bf_pk(p[8 * ks + 0], p[8 * ks + 1]), f2bf_pk(p[8 * ks + 2], p[8 * ks + 3]),
                     f2bf_pk(p[8 * ks + 4], p[8 * ks + 5]), f2bf_pk(p[8 * ks + 6], p[8 * ks + 7]));
    return f.v;
}
// a transposed 32 x 32 result tile (lane: row `row`, columns col0 + rowmap(r, h2)) -> four 8-byte pieces per lane
__device__ __forceinline__ void stage_rows(bf16_t* img, int ld, int row, int col0, int h2, const float (&v)[16]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        uint2 t;
        t.x = f2bf_pk(v[4 * c + 0], v[4 * c + 1]);
        t.y = f2bf_pk(v[4 * c + 2], v[4 * c + 3]);
        *reinterpret_cast<uint2*>(&img[row * ld + col0 + 8 * c + 4 * h2]) = t;
    }
}
// the lane's 16 values of such a tile, read back (bf16 -> fp32)
__device__ __forceinline__ void load_rows(const bf16_t* img, int ld, int row, int col0, int h2, float (&v)[16]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint2 t = *reinterpret_cast<const uint2*>(&img[row * ld + col0 + 8 * c + 4 * h2]);
        v[4 * c + 0] = __uint_as_float(t.x << 16); v[4 * c + 1] = __uint_as_float(t.x & 0xffff0000u);
        v[4 * c + 2] = __uint_as_float(t.y << 16); v[4 * c + 3] = __uint_as_float(t.y & 0xffff0000u);
    }
}
// rows [0, S) x `cols` columns of an image -> global rows (16 bytes per lane, whole workgroup)
__device__ __forceinline__ void store_image(bf16_t* dst, long long ld_dst, const bf16_t* img, int ld, int col0, int S, int cols) {
    const int cpr = cols / 8;
    for (int idx = threadIdx.x; idx < S * cpr; idx += 512) {
        const int r = idx / cpr, c = idx % cpr;
        *reinterpret_cast<uint4*>(dst + (long long)r * ld_dst + 8 * c) = *reinterpret_cast<const uint4*>(img + r * ld + col0 + 8 * c);
    }
}
// hidden image [S][512] -> global rows with fragment-ordered columns (position p holds unit (p & ~12) | ((p & 4) << 1) |
// ((p & 8) >> 1), csrc/ffn_fused.hip frag_pos): the 16-byte piece q of a row = the 8-byte granules at columns
// 16 (q >> 1) + 4 (q & 1) and + 8
__device__ __forceinline__ void store_image_frag512(bf16_t* dst, const bf16_t* img, int ld, int S) {
    for (int idx = threadIdx.x; idx < S * 64; idx += 512) {
        const int r = idx >> 6, q = idx & 63;
        const int c0 = 16 * (q >> 1) + 4 * (q & 1);
        const uint2 lo = *reinterpret_cast<const uint2*>(img + r * ld + c0);
        const uint2 hi = *reinterpret_cast<const uint2*>(img + r * ld + c0 + 8);
        *reinterpret_cast<uint4*>(dst + (long long)r * GF + 8 * q) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
}
// global rows -> image (rows >= S: a copy of row S - 1, finite values that are computed on but never stored)
__device__ __forceinline__ void load_image(bf16_t* img, int ld, int col0, const bf16_t* src, long long ld_src, int S, int cols) {
    const int cpr = cols / 8;
    for (int idx = threadIdx.x; idx < 32 * cpr; idx += 512) {
        const int r = idx / cpr, c = idx % cpr;
        *reinterpret_cast<uint4*>(img + r * ld + col0 + 8 * c) =
            *reinterpret_cast<const uint4*>(src + (long long)min(r, S - 1) * ld_src + 8 * c);
    }
}
// LDS-only workgroup barrier: the weight prefetch (global loads) stays in flight across it
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// drop_make on a seed that is already in a register (the kernels read it once, BEFORE the weight stream starts)
__device__ __forceinline__ DropCtx drop_make_v(float p, bool has_seed, uint64_t seed, uint32_t site) {
    DropCtx c;
    c.on = (p > 0.f) && has_seed;
    if (c.on) {
        c.s0 = dsvg_hash32((uint32_t)seed ^ (site * 0x9e3779b1u));
        c.s1 = dsvg_hash32((uint32_t)(seed >> 32) + site * 0x85ebca77u + 0x165667b1u);
        uint32_t t = (uint32_t)(p * 65536.f + 0.5f);
        c.thresh = t > 65535u ? 65535u : t;
        c.scale = 65536.f / (float)(65536u - c.thresh);
    } else {
        c.s0 = c.s1 = 0; c.thresh = 0; c.scale = 1.f;
    }
    return c;
}

// multipliers of the 4 elements idx8 + 4 h2 .. + 3 of an aligned group of 8 (the half of drop_mult8's group this lane owns in
// a transposed result tile: words 2 h2 and 2 h2 + 1 of the group - the other two are the partner lane's)
__device__ __forceinline__ void drop_mult4(const DropCtx& c, uint64_t idx8, int h2, float (&m)[4]) {
    if (!c.on) {
        m[0] = m[1] = m[2] = m[3] = 1.f;
        return;
    }
    const uint32_t h = drop_group(c, idx8 >> 3);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t w = drop_word(h, 2 * h2 + i);
        m[2 * i] = (w & 0xffffu) < c.thresh ? 0.f : c.scale;
        m[2 * i + 1] = (w >> 16) < c.thresh ? 0.f : c.scale;
    }
}

// The layer's weight image (1 MiB per direction) is cold in this XCD's L2 when the launch starts - inside a training step every
// layer's image is touched once per step - and every workgroup of the XCD walks it in lockstep from the same address on: each
// wave's prefetch ring (PF KiB in flight) then runs at the latency of an HBM miss all the way.  So the workgroups of an XCD
// (block b runs on XCD b % 8) request the WHOLE image up front, 64 KiB each: LDS-DMA into a 1 KiB dump area per wave (no
// registers, nothing to wait for: the bytes are never read - the point is the line in L2).  warm = 0: off (A/B knob DSVG_GS_WARM)
constexpr int GS_WARM_LDS = 8 * 1024;
__device__ __forceinline__ void gs_warm_l2(const bf16_t* img, int wave, int lane, uint32_t lds_dump, int warm) {
    if (!warm) return;
    const uint32_t slice = ((uint32_t)blockIdx.x >> 3) & 15u;
    // per-lane addresses (the saddr form wants a provably wave-uniform base: hipcc hands it VGPRs here)
    const char* b0 = reinterpret_cast<const char*>(img) + ((size_t)slice * 64 + (size_t)wave * 8) * 1024 + (size_t)lane * 16;
    const char* b1 = b0 + 1024; const char* b2 = b0 + 2048; const char* b3 = b0 + 3072;
    const char* b4 = b0 + 4096; const char* b5 = b0 + 5120; const char* b6 = b0 + 6144; const char* b7 = b0 + 7168;
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %9\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "global_load_lds_dwordx4 %2, off\n\t"
        "global_load_lds_dwordx4 %3, off\n\t"
        "global_load_lds_dwordx4 %4, off\n\t"
        "global_load_lds_dwordx4 %5, off\n\t"
        "global_load_lds_dwordx4 %6, off\n\t"
        "global_load_lds_dwordx4 %7, off\n\t"
        "global_load_lds_dwordx4 %8, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7), "s"(lds_dump)
        : "memory");
}

// the per-wave weight stream: fragment i of the layer is `ring[i % PF]` once `take(i)` has been called in order
// PF: prefetch distance of the weight stream (fragments = KiB in flight per wave)
template <int PF>
struct WStream {
    const char* base;       // this lane's 16 bytes of fragment 0
    uint4 ring[PF];
    static constexpr int DEPTH = PF;
    __device__ __forceinline__ void start(const bf16_t* img, int wave, int lane) {
        base = reinterpret_cast<const char*>(img) + ((size_t)wave * GS_FRAGS) * FRAG + lane * 16;
#pragma unroll
        for (int i = 0; i < PF; ++i) ring[i] = *reinterpret_cast<const uint4*>(base + (size_t)i * FRAG);
    }
};
// fragment I (compile-time) as an MFMA operand, and the refill of its ring slot with fragment I + PF
#define GS_TAKE(ws, I, dst)                                                                                      \
    do {                                                                                                         \
        (dst).u = (ws).ring[(I) % PF];                                                                           \
        if ((I) + PF < GS_FRAGS) (ws).ring[(I) % PF] = *reinterpret_cast<const uint4*>((ws).base + (size_t)((I) + PF) * FRAG); \
    } while (0)        /* (PF: the enclosing kernel's template parameter) */

struct GsFwdArgs {
    const bf16_t* x; const bf16_t* img;
    const float* in_bias; const float* out_bias; const float* b1; const float* b2;
    const float* g1; const float* be1; const float* g2; const float* be2;
    const uint64_t* key_mask; const bf16_t* gadd; const uint64_t* seed;
    long long gadd_ld;      // row stride (elements) of gadd
    bf16_t* x2;
    float* mean1; float* rstd1; bf16_t* xn1; bf16_t* qkv; bf16_t* ao; bf16_t* x1;
    float* mean2; float* rstd2; bf16_t* xn2; bf16_t* h;
    int n_seq, S;
    int per;                // whole sequences per tile (<= 32 / S; fewer: see gs_per)
    int warm;               // request the whole weight image into L2 up front (gs_warm_l2)
    float eps, scale, drop_p;
    uint32_t site0;
    long long seq_base;     // the launch covers sequences seq_base .. seq_base + n_seq - 1 of a longer buffer: the pointers are
                            // those of its first row, the dropout draws are indexed from the buffer's first row / sequence
    int ffn_format;         // training outputs of the FFN half as csrc/ffn_fused.hip's backward reads them: xn2 = the
                            // affine-free (x1 - mean2) rstd2, h with fragment-ordered columns (frag_pos)
    unsigned long long* dbg;    // development probe (dsvg_gs_debug_clock): 8 s_memtime stamps per wave, or nullptr
};

// stamp `slot` of this wave: kernel start, LayerNorm 1 done, B1 (in_proj + attention), B2c (out_proj + LayerNorm 2), B3
// (linear1), B4 (linear2), stores issued
__device__ __forceinline__ void gs_stamp(unsigned long long* d, int slot) {
    if (d && (threadIdx.x & 63) == 0)
        d[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + slot] = __builtin_amdgcn_s_memtime();
}

// LDS layout of the forward kernel (bytes)
constexpr int F_XN = 0;                                 // [32][LDX]  LN1(x); later LN2(x1); at the end x2
constexpr int F_AO = F_XN + 32 * LDX * 2;               // [32][LDX]  head outputs; later x1
constexpr int F_QKV = F_AO + 32 * LDX * 2;              // [32][LDQ]  q|k|v;  later the hidden activations [32][LDH]
constexpr int F_SMALL = F_QKV + 32 * LDQ * 2;           // in_bias 768 | out_bias 256 | b1 512 | b2 256 | g1 be1 g2 be2 4 x 256
constexpr int F_STAT = F_SMALL + (768 + 256 + 512 + 256 + 1024) * 4;    // [2][8][32] row-statistic partials
constexpr int F_LDS = F_STAT + 2 * 8 * 32 * 4;

// row statistics of a [32][256] fp32 tile spread as (lane: row li, 16 columns of block w) over the 8 waves: returns the
// row sum of `v` over all 256 columns (one LDS round)
__device__ __forceinline__ float row_total(float part, float* stat, int wave, int li, int h2) {
    part += __shfl_xor(part, 32, 64);
    if (h2 == 0) stat[wave * 32 + li] = part;
    lds_barrier();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) t += stat[w * 32 + li];
    return t;
}

// One layer of a tile.  `first`: the tile's rows come from a.x; otherwise (layers 1 .. of a STACK launch, gs_stack_fwd_kernel)
// they are what the previous layer left on chip - its x2 image in XN and, for the residual, the same values at this lane's
// result positions in `xc` (bf16-rounded, as a.x would hold them) - so a.x is not read at all.  On return xc holds this layer's x2.
template <bool TRAIN, int PF>
__device__ __forceinline__ void gs_fwd_layer(const GsFwdArgs& a, char* smem, const bool first, float (&xc)[16],
                                             const bf16_t* warm_next) {
    gs_stamp(a.dbg, 0);
    bf16_t* XN = reinterpret_cast<bf16_t*>(smem + F_XN);
    bf16_t* AO = reinterpret_cast<bf16_t*>(smem + F_AO);
    bf16_t* QKV = reinterpret_cast<bf16_t*>(smem + F_QKV);
    bf16_t* HI = QKV;
    bf16_t* XH = QKV + (32 * LDQ - 32 * GD);            // [32][256] behind the hidden image (32 * LDH elements), ffn_format only
    float* sbin = reinterpret_cast<float*>(smem + F_SMALL);
    float* sbo = sbin + 768;
    float* sb1 = sbo + 256;
    float* sb2 = sb1 + 512;
    float* sg1 = sb2 + 256;
    float* sbe1 = sg1 + 256;
    float* sg2 = sbe1 + 256;
    float* sbe2 = sg2 + 256;
    float* stat = reinterpret_cast<float*>(smem + F_STAT);

    // (opaque per call: inside a stack launch's layer loop nothing derived from the thread index is hoisted out of the loop and
    // kept in registers across the whole layer)
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int tid = tid_;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, h2 = lane >> 5;
    const int Smax = a.S;
    const int per = a.per;                              // whole sequences per tile
    const int s_first = blockIdx.x * per;
    const int n_in = min(per, a.n_seq - s_first);
    const int S = n_in * Smax;                          // live rows of this tile
    const long long row0 = (long long)s_first * Smax;

    // the tile's own loads go out BEFORE the weight stream starts: loads return in order, so a value issued behind the
    // prefetch ring could only be waited for together with the whole ring
    const int r0 = wave * 4 + (lane >> 4), c00 = (lane & 15) * 16;
    const long long gr0 = row0 + min(r0, S - 1);
    uint4 ra, rb;
    if (first) {
        ra = *reinterpret_cast<const uint4*>(a.x + gr0 * GD + c00);
        rb = *reinterpret_cast<const uint4*>(a.x + gr0 * GD + c00 + 8);
    } else {
        // (the previous layer's barrier B4 is behind every wave: the image is complete; phase 0 overwrites it behind a barrier)
        ra = *reinterpret_cast<const uint4*>(&XN[r0 * LDX + c00]);
        rb = *reinterpret_cast<const uint4*>(&XN[r0 * LDX + c00 + 8]);
    }
    const bool has_seed = a.seed != nullptr && a.drop_p > 0.f;
    const uint64_t seedv = has_seed ? *a.seed : 0ull;
    for (int i = tid; i < 768; i += 512) sbin[i] = a.in_bias[i];
    sb1[tid] = a.b1[tid];
    if (tid < 256) {
        sbo[tid] = a.out_bias[tid]; sb2[tid] = a.b2[tid];
        sg1[tid] = a.g1[tid]; sbe1[tid] = a.be1[tid]; sg2[tid] = a.g2[tid]; sbe2[tid] = a.be2[tid];
    }
    const uint32_t dump = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)DSVG_LDS_PTR(smem) + F_LDS + wave * 1024);
    // warm (DSVG_GS_WARM): 1 = this layer's image up front - every layer of a stack launch, like the per-layer launches; 2 (stack launches,
    // A/B) = layer 0's up front, the NEXT layer's before every layer's last phase; 3 = both
    if ((a.warm & 1) || (first && a.warm)) gs_warm_l2(a.img, wave, lane, dump, 1);
    WStream<PF> ws;
    ws.start(a.img, wave, lane);

    // ---- phase 0: LayerNorm 1, 16 lanes per row (wave w: rows 4 w .. 4 w + 3), 16 columns per lane -----------------------
    {
        const int r = r0, c0 = c00;
        const long long gr = gr0;
        float v[16];
        {
            float t[8];
            unpack8(ra, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = t[e];
            unpack8(rb, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[8 + e] = t[e];
        }
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) s += v[e];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s * (1.f / GD);
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float d = v[e] - mean; ss += d * d; }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float rstd = rsqrtf(ss * (1.f / GD) + a.eps);
        lds_barrier();              // gamma / beta (and the biases) staged
        float y[8], z[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            y[e] = (v[e] - mean) * rstd * sg1[c0 + e] + sbe1[c0 + e];
            z[e] = (v[8 + e] - mean) * rstd * sg1[c0 + 8 + e] + sbe1[c0 + 8 + e];
        }
        const uint4 pa = pack8(y), pb = pack8(z);
        *reinterpret_cast<uint4*>(&XN[r * LDX + c0]) = pa;
        *reinterpret_cast<uint4*>(&XN[r * LDX + c0 + 8]) = pb;
        if (TRAIN && r < S) {
            *reinterpret_cast<uint4*>(a.xn1 + gr * GD + c0) = pa;
            *reinterpret_cast<uint4*>(a.xn1 + gr * GD + c0 + 8) = pb;
            if ((lane & 15) == 0) { a.mean1[gr] = mean; a.rstd1[gr] = rstd; }
        }
    }
    lds_barrier();
    gs_stamp(a.dbg, 1);

    // ---- the lane's attention row: its sequence, the first row of that sequence inside the tile, visible keys -------------
    const int qi = min(li / Smax, n_in - 1);
    const int my_seq = s_first + qi, my_start = qi * Smax;
    const uint32_t seqbits = (uint32_t)((1ull << Smax) - 1ull);
    const uint32_t km = (((a.key_mask ? (uint32_t)a.key_mask[my_seq] : ~0u) & seqbits) << my_start);
    const DropCtx dp = drop_make_v(a.drop_p, has_seed, seedv, a.site0);
    const DropCtx dr1 = drop_make_v(a.drop_p, has_seed, seedv, a.site0 + 1);
    const DropCtx dg = drop_make_v(a.drop_p, has_seed, seedv, a.site0 + 2);
    const DropCtx dh = drop_make_v(a.drop_p, has_seed, seedv, a.site0 + 3);
    const DropCtx dr2 = drop_make_v(a.drop_p, has_seed, seedv, a.site0 + 4);
    const long long m = row0 + li;                      // this lane's row (result tiles: lane = row li)
    const uint64_t dseq = (uint64_t)(my_seq + a.seq_base);              // ... and their indices for the dropout draws
    const uint64_t drow = (uint64_t)(m + a.seq_base * Smax);
    const bool live = li < S;
    const long long mrow = row0 + min(li, S - 1);
    const int qc = wave * 32, kc = GD + wave * 32, vc = 2 * GD + wave * 32;

    // ---- phase 1: in_proj of head `wave` (q, k, v column blocks) + attention ----------------------------------------------
    {
        floatx16 qa, ka, va;
#pragma unroll
        for (int r = 0; r < 16; ++r) { qa[r] = 0.f; ka[r] = 0.f; va[r] = 0.f; }
        bf16x8 b = row_frag(XN, LDX, li, 0, 0, h2);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const bf16x8 bn = ks + 1 < 16 ? row_frag(XN, LDX, li, 0, ks + 1, h2) : b;
            Frag8 w0, w1, w2;
            GS_TAKE(ws, 3 * ks + 0, w0);
            GS_TAKE(ws, 3 * ks + 1, w1);
            GS_TAKE(ws, 3 * ks + 2, w2);
            qa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0.v, b, qa, 0, 0, 0);
            ka = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1.v, b, ka, 0, 0, 0);
            va = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2.v, b, va, 0, 0, 0);
            b = bn;
            __builtin_amdgcn_sched_barrier(0);
        }
        float t[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = qa[r] + sbin[qc + rowmap(r, h2)];
        stage_rows(QKV, LDQ, li, qc, h2, t);
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = ka[r] + sbin[kc + rowmap(r, h2)];
        stage_rows(QKV, LDQ, li, kc, h2, t);
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = va[r] + sbin[vc + rowmap(r, h2)];
        stage_rows(QKV, LDQ, li, vc, h2, t);
    }
    // the head's q | k | v slabs were written and are read by this wave only (in-order LDS): no workgroup barrier
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
        floatx16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int step = 0; step < 2; ++step)
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(QKV, LDQ, li, kc, step, h2), row_frag(QKV, LDQ, li, qc, step, h2),
                                                         st, 0, 0, 0);
        // st[r] = q_li . k_key(r, h2): softmax over the keys of the lane's own sequence that its key mask lets through
        float p[16];
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = ((km >> rowmap(r, h2)) & 1u) ? st[r] * a.scale : -INFINITY;
            mx = fmaxf(mx, p[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = (p[r] == -INFINITY) ? 0.f : __expf(p[r] - mx);
            l += p[r];
        }
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.f / l;
        // dropout row of (sequence s, head h, query i) = (s H + h) Smax + i; query and key counted inside the sequence
        const uint32_t hrow = attn_drop_row(dp, (dseq * GH + wave) * Smax + (li - my_start), 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = p[r] * inv * attn_drop_key(dp, hrow, (uint32_t)(rowmap(r, h2) - my_start));
        if (!live) {
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = 0.f;
        }
        floatx16 ot;
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            ot = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(QKV, LDQ, vc, ks, lane), pack_regs(p, ks), ot, 0, 0, 0);
        float t[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = ot[r];
        stage_rows(AO, LDX, li, qc, h2, t);             // ot[r] = O[row li][dim rowmap(r, h2)] of head `wave`
    }
    lds_barrier();                                       // B1: q|k|v and the head outputs of all heads are in LDS
    gs_stamp(a.dbg, 2);
    if (TRAIN) {
        store_image(a.qkv + row0 * (3 * GD), 3 * GD, QKV, LDQ, 0, S, 3 * GD);
        store_image(a.ao + row0 * GD, GD, AO, LDX, 0, S, GD);
    }

    // ---- phase 2: out_proj column block `wave`, bias, dropout, residual (+ the per-sequence conditioning row), LayerNorm 2 --
    float x1v[16];
    {
        // the residual row pieces (L2-hot: phase 0 read them) and the conditioning row, issued ahead of the product
        uint2 xr[4], gr4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (first) xr[c] = *reinterpret_cast<const uint2*>(a.x + mrow * GD + qc + 8 * c + 4 * h2);
            if (a.gadd) gr4[c] = *reinterpret_cast<const uint2*>(a.gadd + (long long)my_seq * a.gadd_ld + qc + 8 * c + 4 * h2);
        }
        floatx16 ya;
#pragma unroll
        for (int r = 0; r < 16; ++r) ya[r] = 0.f;
        bf16x8 b = row_frag(AO, LDX, li, 0, 0, h2);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const bf16x8 bn = ks + 1 < 16 ? row_frag(AO, LDX, li, 0, ks + 1, h2) : b;
            Frag8 w0;
            GS_TAKE(ws, 48 + ks, w0);
            ya = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0.v, b, ya, 0, 0, 0);
            b = bn;
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // columns qc + 8 c + 4 h2 .. + 3: half of the aligned group of 8 the standard draws are made for
            float dm[4], gm[4];
            drop_mult4(dr1, drow * GD + qc + 8 * c, h2, dm);
            if (a.gadd) drop_mult4(dg, dseq * GD + qc + 8 * c, h2, gm);
            float xv[4];
            if (first) {
                xv[0] = __uint_as_float(xr[c].x << 16); xv[1] = __uint_as_float(xr[c].x & 0xffff0000u);
                xv[2] = __uint_as_float(xr[c].y << 16); xv[3] = __uint_as_float(xr[c].y & 0xffff0000u);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) xv[e] = xc[4 * c + e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = (ya[4 * c + e] + sbo[qc + 8 * c + 4 * h2 + e]) * dm[e] + xv[e];
                if (a.gadd) {
                    const uint32_t gw = e < 2 ? gr4[c].x : gr4[c].y;
                    const float gv = (e & 1) ? __uint_as_float(gw & 0xffff0000u) : __uint_as_float(gw << 16);
                    v += gv * gm[e];
                }
                // (the value every later stage sees is the stored bf16 one, as on the unfused path)
                x1v[4 * c + e] = bf2f(f2bf(v));
            }
        }
    }
    float mean2, rstd2;
    {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += x1v[r];
        mean2 = row_total(s, stat, wave, li, h2) * (1.f / GD);                 // (barrier B2a inside)
        float ss = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = x1v[r] - mean2; ss += d * d; }
        rstd2 = rsqrtf(row_total(ss, stat + 256, wave, li, h2) * (1.f / GD) + a.eps);      // (B2b)
        float t[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int col = qc + rowmap(r, h2);
            t[r] = (x1v[r] - mean2) * rstd2 * sg2[col] + sbe2[col];
        }
        // every wave is past B2a: nobody reads XN (in_proj) or AO (out_proj) any more
        stage_rows(XN, LDX, li, qc, h2, t);
        if (TRAIN) stage_rows(AO, LDX, li, qc, h2, x1v);
        if (TRAIN && a.ffn_format) {
            // the affine-free rows for the fused FFN backward, staged behind the hidden image's extent in the (dead) q|k|v
            // region: [32][256], unpadded - written once, its bank conflicts do not matter
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = (x1v[r] - mean2) * rstd2;
            stage_rows(XH, GD, li, qc, h2, t);
        }
    }
    lds_barrier();                                       // B2c: LN2(x1) (and x1) complete
    gs_stamp(a.dbg, 3);
    if (TRAIN) {
        if (a.ffn_format) store_image(a.xn2 + row0 * GD, GD, XH, GD, 0, S, GD);
        else store_image(a.xn2 + row0 * GD, GD, XN, LDX, 0, S, GD);
        store_image(a.x1 + row0 * GD, GD, AO, LDX, 0, S, GD);
        if (wave == 0 && h2 == 0 && live) { a.mean2[m] = mean2; a.rstd2[m] = rstd2; }
    }

    // ---- phase 3: linear1 hidden column blocks 2 w, 2 w + 1, bias, ReLU, dropout -> hidden image -------------------------------
    {
        floatx16 h0, h1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { h0[r] = 0.f; h1[r] = 0.f; }
        bf16x8 b = row_frag(XN, LDX, li, 0, 0, h2);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const bf16x8 bn = ks + 1 < 16 ? row_frag(XN, LDX, li, 0, ks + 1, h2) : b;
            Frag8 w0, w1;
            GS_TAKE(ws, 64 + 2 * ks + 0, w0);
            GS_TAKE(ws, 64 + 2 * ks + 1, w1);
            h0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0.v, b, h0, 0, 0, 0);
            h1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1.v, b, h1, 0, 0, 0);
            b = bn;
            __builtin_amdgcn_sched_barrier(0);
        }
        // (the q|k|v image is dead: its last readers - the training stores behind B1 - finished before their wave reached B2a)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int hc = 64 * wave + 32 * jj;
            float t[16];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float dm[4];
                drop_mult4(dh, drow * GF + hc + 8 * c, h2, dm);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pre = (jj ? h1[4 * c + e] : h0[4 * c + e]) + sb1[hc + 8 * c + 4 * h2 + e];
                    t[4 * c + e] = fmaxf(pre, 0.f) * dm[e];
                }
            }
            stage_rows(HI, LDH, li, hc, h2, t);
        }
    }
    lds_barrier();                                       // B3: the hidden activations of the tile are in LDS
    gs_stamp(a.dbg, 4);
    if (warm_next && (a.warm & 2)) gs_warm_l2(warm_next, wave, lane, dump, 1);
    if (TRAIN) {
        if (a.ffn_format) store_image_frag512(a.h + row0 * GF, HI, LDH, S);
        else store_image(a.h + row0 * GF, GF, HI, LDH, 0, S, GF);
    }

    // ---- phase 4: linear2 column block `wave`, bias, dropout, residual -> x2 ----------------------------------------------------
    {
        floatx16 ya;
#pragma unroll
        for (int r = 0; r < 16; ++r) ya[r] = 0.f;
        bf16x8 b = row_frag(HI, LDH, li, 0, 0, h2);
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
            const bf16x8 bn = ks + 1 < 32 ? row_frag(HI, LDH, li, 0, ks + 1, h2) : b;
            Frag8 w0;
            GS_TAKE(ws, 96 + ks, w0);
            ya = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0.v, b, ya, 0, 0, 0);
            b = bn;
            __builtin_amdgcn_sched_barrier(0);
        }
        float t[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float dm[4];
            drop_mult4(dr2, drow * GD + qc + 8 * c, h2, dm);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                t[4 * c + e] = (ya[4 * c + e] + sb2[qc + 8 * c + 4 * h2 + e]) * dm[e] + x1v[4 * c + e];
        }
        // every wave is past B3: nobody reads XN (linear1) any more
        stage_rows(XN, LDX, li, qc, h2, t);
#pragma unroll
        for (int r = 0; r < 16; ++r) xc[r] = bf2f(f2bf(t[r]));          // (what the stored x2 holds)
    }
    lds_barrier();                                       // B4
    gs_stamp(a.dbg, 5);
    if (a.x2) store_image(a.x2 + row0 * GD, GD, XN, LDX, 0, S, GD);     // (NULL: an inner layer of an inference stack launch)
    gs_stamp(a.dbg, 6);
}

// Block l of an array of argument structs that starts the kernel's argument list.  (A run-time index into the by-value argument
// would move it to scratch memory: the words are read from the kernel argument segment itself - scalar loads, constant memory.)
typedef const __attribute__((address_space(4))) unsigned long long* gs_kargs_t;
template <typename T>
__device__ __forceinline__ T gs_karg(int l) {
    static_assert(sizeof(T) % 8 == 0, "whole 8-byte words");
    constexpr int W = sizeof(T) / 8;
    gs_kargs_t kw = (gs_kargs_t)__builtin_amdgcn_kernarg_segment_ptr() + (size_t)l * W;
    union U { T a; unsigned long long w[W]; __device__ U() {} } u;
#pragma unroll
    for (int i = 0; i < W; ++i) u.w[i] = kw[i];
    return u.a;
}
// A pointer that was read as a plain word is a GENERIC pointer to the compiler: every access through it becomes a flat_load /
// flat_store, which counts on vmcnt AND lgkmcnt and may return out of order with LDS traffic - the counted waits of the weight
// prefetch ring degrade to "wait for everything" (measured: the stack launch 10 % slower than its layers launched one by one).
// Rebuilt from the word as an address-space-1 pointer it is what a by-value kernel argument is implicitly: a global pointer.
template <typename T>
__device__ __forceinline__ T* gs_global(T* p) {
    return (T*)(T __attribute__((address_space(1)))*)(unsigned long long)p;
}
__device__ __forceinline__ const bf16_t* gs_karg_ptr(size_t byte_off) {
    gs_kargs_t kw = (gs_kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
    return gs_global(reinterpret_cast<const bf16_t*>(kw[byte_off / 8]));
}
__device__ __forceinline__ GsFwdArgs gs_globalize(GsFwdArgs a) {
#define G(f) a.f = gs_global(a.f)
    G(x); G(img); G(in_bias); G(out_bias); G(b1); G(b2); G(g1); G(be1); G(g2); G(be2); G(key_mask); G(gadd); G(seed);
    G(x2); G(mean1); G(rstd1); G(xn1); G(qkv); G(ao); G(x1); G(mean2); G(rstd2); G(xn2); G(h); G(dbg);
#undef G
    return a;
}

template <bool TRAIN, int PF>
__global__ __launch_bounds__(512, 2) void gs_layer_fwd_kernel(const GsFwdArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    float xc[16];
    gs_fwd_layer<TRAIN, PF>(a, smem, true, xc, nullptr);
}

// A whole STACK of layers in one launch (round 6): the tiles are independent across the layers (a tile holds whole sequences), so
// a workgroup carries its 32 rows through all of them - no launch ramp / drain per layer, the rows never come back from memory,
// and layer l + 1's weight image is requested into L2 while layer l computes.  Everything a layer stores for the backward pass
// is stored as by the per-layer launches: bit-identical results (tests/test_kernels_gpu.py).
constexpr int GS_STACK_MAX = 4;
struct GsFwdStackArgs {
    GsFwdArgs layer[GS_STACK_MAX];
    int n_layers;
};
template <bool TRAIN, int PF>
__global__ __launch_bounds__(512, 2) void gs_stack_fwd_kernel(const GsFwdStackArgs A) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    // (a run-time index into the by-value argument would move it to scratch memory: the layer's block is read from the kernel
    // argument segment itself - scalar loads from constant memory)
    const int n = A.n_layers;
    float xc[16];
#pragma unroll 1
    for (int l = 0; l < n; ++l) {
        const GsFwdArgs a = gs_globalize(gs_karg<GsFwdArgs>(l));
        const bf16_t* next = l + 1 < n ? gs_karg_ptr((size_t)(l + 1) * sizeof(GsFwdArgs) + offsetof(GsFwdArgs, img)) : nullptr;
        gs_fwd_layer<TRAIN, PF>(a, smem, l == 0, xc, next);
        // (the next layer's first LDS writes - parameter staging - touch nothing store_image reads; its phase 0 writes XN behind a barrier)
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// backward: from dx2 = dL/dx2 and what the forward pass saved to dx = dL/dx, the token-major operands of the layer's four
// weight-gradient GEMMs (dym = drop-mask(dx2), dpre, dx1m = drop-mask(dx1), dqkv; the other operands - h, LN2(x1), the head
// outputs, LN1(x) - were stored by the forward pass), dx1 (for the conditioning row's gradient) and per-tile partial sums of
// the four LayerNorm parameter gradients (column sums over the tile's rows by MFMA against ones).
// ---------------------------------------------------------------------------------------------------------------------
struct GsBwdArgs {
    const bf16_t* dx2; const bf16_t* img;
    const bf16_t* x; const float* mean1; const float* rstd1; const bf16_t* qkv;
    const bf16_t* x1; const float* mean2; const float* rstd2; const bf16_t* h;
    const float* g1; const float* g2;
    const uint64_t* key_mask; const uint64_t* seed;
    bf16_t* dx; bf16_t* dx1; bf16_t* dym; bf16_t* dpre; bf16_t* dx1m; bf16_t* dqkv;
    bf16_t* dg;             // [n_seq][dg_ld] or NULL: the per-sequence term's gradient (what dsvg_bcast_add_bwd makes of dx1)
    long long dg_ld;        // row stride (elements) of dg: 256, or the stack's concatenated buffer (layer l = column block l)
    float* ln_part;         // [tiles][4][256]: dgamma2, dbeta2, dgamma1, dbeta1
    int n_seq, S;
    int per;                // whole sequences per tile (see gs_per)
    int warm;               // request the whole weight image into L2 up front (gs_warm_l2)
    float scale, drop_p;
    uint32_t site0;
};

constexpr int B_A0 = 0;                                 // [32][LDX]  dx2; later dx1m; later g1 * xh1
constexpr int B_A1 = B_A0 + 32 * LDX * 2;               // [32][LDX]  dym; later g2; later g1
constexpr int B_DO = B_A1 + 32 * LDX * 2;               // [32][LDX]  g2 * xh2; later the head-output gradients
constexpr int B_HI = B_DO + 32 * LDX * 2;               // [32][LDH]  dpre; later dx1 / dx as [32][LDX]
constexpr int B_QKV = B_HI + 32 * LDH * 2;              // [32][LDQ]  q|k|v; later dq|dk|dv
constexpr int B_SMALL = B_QKV + 32 * LDQ * 2;           // gamma1 | gamma2
constexpr int B_STAT = B_SMALL + 2 * 256 * 4;           // [2][8][32] row sums of the LayerNorm backward
constexpr int B_ASTAT = B_STAT + 2 * 8 * 32 * 4;        // [8][96]    lse, D, dropout row hash per head and query
constexpr int B_LDS = B_ASTAT + 8 * 96 * 4;

// LayerNorm backward of a row spread over the 8 waves (lane: row li, columns col0 + rowmap(r, h2)):
//   gh = g * gamma;  out = res + rstd * (gh - mean(gh) - xh * mean(gh * xh));  t1 = g * xh, t2 = g (parameter-gradient addends)
__device__ __forceinline__ void ln_bwd_rows(const float (&g)[16], const float (&xv)[16], float mean, float rstd,
                                            const float* gamma, int col0, int h2, const float (&res)[16], float* stat,
                                            int wave, int li, bool live, float (&out)[16], float (&t1)[16], float (&t2)[16]) {
    float gh[16], xh[16];
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        xh[r] = (xv[r] - mean) * rstd;
        gh[r] = g[r] * gamma[col0 + rowmap(r, h2)];
        sa += gh[r];
        sb = fmaf(gh[r], xh[r], sb);
    }
    sa += __shfl_xor(sa, 32, 64);
    sb += __shfl_xor(sb, 32, 64);
    if (h2 == 0) { stat[wave * 32 + li] = sa; stat[256 + wave * 32 + li] = sb; }
    lds_barrier();
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { c1 += stat[w * 32 + li]; c2 += stat[256 + w * 32 + li]; }
    c1 *= (1.f / GD);
    c2 *= (1.f / GD);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        out[r] = res[r] + rstd * (gh[r] - c1 - xh[r] * c2);
        t1[r] = live ? g[r] * xh[r] : 0.f;
        t2[r] = live ? g[r] : 0.f;
    }
}

// column sums over the 32 rows of two staged images (columns col0 .. col0 + 31) -> part[0 .. 255] / part[256 .. 511]
__device__ __forceinline__ void col_sums(const bf16_t* imgA, const bf16_t* imgB, int col0, int lane, float* part) {
    Frag8 ones;
    ones.u = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    floatx16 sa, sb;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sa[r] = 0.f; sb[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(imgA, LDX, col0, ks, lane), ones.v, sa, 0, 0, 0);
        sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(imgB, LDX, col0, ks, lane), ones.v, sb, 0, 0, 0);
    }
    if ((lane & 31) == 0) {         // every result column holds the same sums: lanes 0 and 32 own 16 features each
        const int h2 = lane >> 5;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            *reinterpret_cast<float4*>(part + col0 + 8 * c + 4 * h2) = make_float4(sa[4 * c], sa[4 * c + 1], sa[4 * c + 2], sa[4 * c + 3]);
            *reinterpret_cast<float4*>(part + 256 + col0 + 8 * c + 4 * h2) = make_float4(sb[4 * c], sb[4 * c + 1], sb[4 * c + 2], sb[4 * c + 3]);
        }
    }
}

__device__ __forceinline__ void unpack4(const uint2 t, float* v) {
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}

// One layer of a tile.  `first`: the incoming gradient comes from a.dx2; otherwise (a STACK launch, gs_stack_bwd_kernel, walking
// the layers from the last to the first) it is the dx image the layer above left in HI.  a.dx may be NULL (nobody reads the
// gradient between two layers of a stack launch).
template <int PF>
__device__ __forceinline__ void gs_bwd_layer(const GsBwdArgs& a, char* smem, const bool first, const bf16_t* warm_next) {
    bf16_t* A0 = reinterpret_cast<bf16_t*>(smem + B_A0);
    bf16_t* A1 = reinterpret_cast<bf16_t*>(smem + B_A1);
    bf16_t* DO = reinterpret_cast<bf16_t*>(smem + B_DO);
    bf16_t* HI = reinterpret_cast<bf16_t*>(smem + B_HI);
    bf16_t* QKV = reinterpret_cast<bf16_t*>(smem + B_QKV);
    float* sg1 = reinterpret_cast<float*>(smem + B_SMALL);
    float* sg2 = sg1 + 256;
    float* stat = reinterpret_cast<float*>(smem + B_STAT);
    float* my_stat_base = reinterpret_cast<float*>(smem + B_ASTAT);

    // (opaque per call: inside a stack launch's layer loop nothing derived from the thread index is hoisted out of the loop and
    // kept in registers across the whole layer)
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int tid = tid_;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* my_stat = my_stat_base + wave * 96;
    const int li = lane & 31, h2 = lane >> 5;
    const int Smax = a.S;
    const int per = a.per;
    const int s_first = blockIdx.x * per;
    const int n_in = min(per, a.n_seq - s_first);
    const int S = n_in * Smax;
    const long long row0 = (long long)s_first * Smax;
    const long long m = row0 + li;
    const bool live = li < S;
    const long long mrow = row0 + min(li, S - 1);
    const int qc = wave * 32, kc = GD + wave * 32, vc = 2 * GD + wave * 32;

    // ---- phase 0: everything the tile reads up front goes out before the weight stream starts ----------------------------------
    const int r0 = wave * 4 + (lane >> 4), c00 = (lane & 15) * 16;
    const long long gr0 = row0 + min(r0, S - 1);
    uint4 ra, rb;
    if (first) {
        ra = *reinterpret_cast<const uint4*>(a.dx2 + gr0 * GD + c00);
        rb = *reinterpret_cast<const uint4*>(a.dx2 + gr0 * GD + c00 + 8);
    } else {
        // (behind the layer above's barrier B4b; HI is next written behind this layer's barrier B0)
        ra = *reinterpret_cast<const uint4*>(&HI[r0 * LDX + c00]);
        rb = *reinterpret_cast<const uint4*>(&HI[r0 * LDX + c00 + 8]);
    }
    // the q|k|v rows of the tile: 32 x 96 pieces of 16 bytes, 6 per thread (named registers: an array lands in scratch memory)
#define GS_QV_SRC(k) (a.qkv + (row0 + min((tid + 512 * (k)) / 96, S - 1)) * (3 * GD) + 8 * ((tid + 512 * (k)) % 96))
    const uint4 qv0 = *reinterpret_cast<const uint4*>(GS_QV_SRC(0)), qv1 = *reinterpret_cast<const uint4*>(GS_QV_SRC(1));
    const uint4 qv2 = *reinterpret_cast<const uint4*>(GS_QV_SRC(2)), qv3 = *reinterpret_cast<const uint4*>(GS_QV_SRC(3));
    const uint4 qv4 = *reinterpret_cast<const uint4*>(GS_QV_SRC(4)), qv5 = *reinterpret_cast<const uint4*>(GS_QV_SRC(5));
#undef GS_QV_SRC
    const bool has_seed = a.seed != nullptr && a.drop_p > 0.f;
    const uint64_t seedv = has_seed ? *a.seed : 0ull;
    const float g1v = tid < 256 ? a.g1[tid] : a.g2[tid - 256];
    const float mean2 = a.mean2[mrow], rstd2 = a.rstd2[mrow], mean1 = a.mean1[mrow], rstd1 = a.rstd1[mrow];
    // hidden activations at this lane's positions of the two hidden column blocks (gate of the ReLU + dropout backward)
    uint2 hq[2][4], x1q[4], xq[4];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            hq[jj][c] = *reinterpret_cast<const uint2*>(a.h + mrow * GF + 64 * wave + 32 * jj + 8 * c + 4 * h2);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        x1q[c] = *reinterpret_cast<const uint2*>(a.x1 + mrow * GD + qc + 8 * c + 4 * h2);
        xq[c] = *reinterpret_cast<const uint2*>(a.x + mrow * GD + qc + 8 * c + 4 * h2);
    }
    const int qi = min(li / Smax, n_in - 1);
    const int my_seq = s_first + qi, my_start = qi * Smax;
    const uint32_t kmask_raw = a.key_mask ? (uint32_t)a.key_mask[my_seq] : ~0u;
    const uint32_t dump = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)DSVG_LDS_PTR(smem) + B_LDS + wave * 1024);
    if ((a.warm & 1) || (first && a.warm)) gs_warm_l2(a.img, wave, lane, dump, 1);
    WStream<PF> ws;
    ws.start(a.img, wave, lane);

    const DropCtx dp = drop_make_v(a.drop_p, has_seed, seedv, a.site0);
    const DropCtx dr1 = drop_make_v(a.drop_p, has_seed, seedv, a.site0 + 1);
    const DropCtx dr2 = drop_make_v(a.drop_p, has_seed, seedv, a.site0 + 4);
    const float inv_keep = (a.drop_p > 0.f && has_seed) ? dr2.scale : 1.f;     // (same p at every site of the layer)
    (tid < 256 ? sg1[tid] : sg2[tid - 256]) = g1v;
    {
        // dx2 -> A0, dym = dx2 * mask(FFN residual site) -> A1 and to memory (the B operand of dh and of dW2)
        float v[8], mm[8];
        *reinterpret_cast<uint4*>(&A0[r0 * LDX + c00]) = ra;
        *reinterpret_cast<uint4*>(&A0[r0 * LDX + c00 + 8]) = rb;
        unpack8(ra, v);
        drop_mult8(dr2, (uint64_t)gr0 * GD + c00, mm);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= mm[e];
        const uint4 pa = pack8(v);
        unpack8(rb, v);
        drop_mult8(dr2, (uint64_t)gr0 * GD + c00 + 8, mm);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= mm[e];
        const uint4 pb = pack8(v);
        *reinterpret_cast<uint4*>(&A1[r0 * LDX + c00]) = pa;
        *reinterpret_cast<uint4*>(&A1[r0 * LDX + c00 + 8]) = pb;
        if (r0 < S) {
            *reinterpret_cast<uint4*>(a.dym + gr0 * GD + c00) = pa;
            *reinterpret_cast<uint4*>(a.dym + gr0 * GD + c00 + 8) = pb;
        }
#define GS_QV_DST(k) (&QKV[((tid + 512 * (k)) / 96) * LDQ + 8 * ((tid + 512 * (k)) % 96)])
        *reinterpret_cast<uint4*>(GS_QV_DST(0)) = qv0; *reinterpret_cast<uint4*>(GS_QV_DST(1)) = qv1;
        *reinterpret_cast<uint4*>(GS_QV_DST(2)) = qv2; *reinterpret_cast<uint4*>(GS_QV_DST(3)) = qv3;
        *reinterpret_cast<uint4*>(GS_QV_DST(4)) = qv4; *reinterpret_cast<uint4*>(GS_QV_DST(5)) = qv5;
#undef GS_QV_DST
    }
    lds_barrier();                                       // B0

    // ---- phase 1: dh = dym . W2 (hidden column blocks 2 w, 2 w + 1), gated by h -> dpre ---------------------------------------
    {
        floatx16 d0, d1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { d0[r] = 0.f; d1[r] = 0.f; }
        bf16x8 b = row_frag(A1, LDX, li, 0, 0, h2);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const bf16x8 bn = ks + 1 < 16 ? row_frag(A1, LDX, li, 0, ks + 1, h2) : b;
            Frag8 w0, w1;
            GS_TAKE(ws, 2 * ks + 0, w0);
            GS_TAKE(ws, 2 * ks + 1, w1);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0.v, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1.v, b, d1, 0, 0, 0);
            b = bn;
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            float t[16];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float hv[4];
                unpack4(hq[jj][c], hv);
#pragma unroll
                for (int e = 0; e < 4; ++e)         // h > 0 <=> the unit passed the ReLU and was kept by the dropout
                    t[4 * c + e] = hv[e] > 0.f ? (jj ? d1[4 * c + e] : d0[4 * c + e]) * inv_keep : 0.f;
            }
            stage_rows(HI, LDH, li, 64 * wave + 32 * jj, h2, t);
        }
    }
    lds_barrier();                                       // B1: dpre of the tile
    store_image(a.dpre + row0 * GF, GF, HI, LDH, 0, S, GF);

    // ---- phase 2: dxn2 = dpre . W1 (column block w), LayerNorm 2 backward, + dx2 -> dx1 ------------------------------------------
    float dx1v[16];
    {
        floatx16 ga;
#pragma unroll
        for (int r = 0; r < 16; ++r) ga[r] = 0.f;
        bf16x8 b = row_frag(HI, LDH, li, 0, 0, h2);
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
            const bf16x8 bn = ks + 1 < 32 ? row_frag(HI, LDH, li, 0, ks + 1, h2) : b;
            Frag8 w0;
            GS_TAKE(ws, 32 + ks, w0);
            ga = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0.v, b, ga, 0, 0, 0);
            b = bn;
            __builtin_amdgcn_sched_barrier(0);
        }
        float g[16], xv[16], res[16], t1[16], t2[16], o[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = bf2f(f2bf(ga[r]));      // (the unfused path stores dxn2 as bf16 between the launches)
#pragma unroll
        for (int c = 0; c < 4; ++c) unpack4(x1q[c], &xv[4 * c]);
        load_rows(A0, LDX, li, qc, h2, res);                          // dx2 at this lane's positions
        ln_bwd_rows(g, xv, mean2, rstd2, sg2, qc, h2, res, stat, wave, li, live, o, t1, t2);       // (barrier B2a inside)
        // every wave is past B2a: the dxn2 products (readers of HI) and the dh products (readers of A1) are complete
        stage_rows(DO, LDX, li, qc, h2, t1);
        stage_rows(A1, LDX, li, qc, h2, t2);
#pragma unroll
        for (int r = 0; r < 16; ++r) dx1v[r] = bf2f(f2bf(o[r]));
        stage_rows(HI, LDX, li, qc, h2, dx1v);
        float t[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float dm[4];
            drop_mult4(dr1, (uint64_t)m * GD + qc + 8 * c, h2, dm);
#pragma unroll
            for (int e = 0; e < 4; ++e) t[4 * c + e] = dx1v[4 * c + e] * dm[e];
        }
        stage_rows(A0, LDX, li, qc, h2, t);             // (this wave's own column block of A0: read above, by this wave only)
    }
    lds_barrier();                                       // B2b: dx1m, dx1 and the LN2 gradient addends of the tile
    float* part = a.ln_part + (size_t)blockIdx.x * 1024;
    col_sums(DO, A1, qc, lane, part);                    // dgamma2 | dbeta2, columns of block w
    store_image(a.dx1m + row0 * GD, GD, A0, LDX, 0, S, GD);
    if (a.dx1) store_image(a.dx1 + row0 * GD, GD, HI, LDX, 0, S, GD);
    if (a.dg) {
        // dg[sequence] = mask(site0 + 2) * sum over the sequence's rows of dx1 - the tile holds whole sequences, so the launch of
        // dsvg_bcast_add_bwd behind this kernel (and, for its sake, the dx1 store) is not needed.  Same summation order as that
        // kernel (rows i = q mod 4 in increasing order, then ((s0 + s1) + s2) + s3) and same draws: bit-identical
        const DropCtx dgc = drop_make_v(a.drop_p, has_seed, seedv, a.site0 + 2);
        for (int idx = tid; idx < n_in * 32; idx += 512) {
            const int sq = idx >> 5, c8 = (idx & 31) * 8;
            float sp[4][8];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) sp[q][e] = 0.f;
            for (int i0 = 0; i0 < Smax; i0 += 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (i0 + q >= Smax) break;
                    float v[8];
                    unpack8(*reinterpret_cast<const uint4*>(&HI[(sq * Smax + i0 + q) * LDX + c8]), v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) sp[q][e] += v[e];
                }
            }
            float sv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) sv[e] = ((sp[0][e] + sp[1][e]) + sp[2][e]) + sp[3][e];
            if (dgc.on) {
                float dm[8];
                drop_mult8(dgc, (uint64_t)(s_first + sq) * GD + c8, dm);
#pragma unroll
                for (int e = 0; e < 8; ++e) sv[e] *= dm[e];
            }
            *reinterpret_cast<uint4*>(a.dg + (long long)(s_first + sq) * a.dg_ld + c8) = pack8(sv);
        }
    }

    // ---- phase 3: dao = dx1m . Wo (the dims of head w) + attention backward of head w -----------------------------------------
    {
        floatx16 oa;
#pragma unroll
        for (int r = 0; r < 16; ++r) oa[r] = 0.f;
        bf16x8 b = row_frag(A0, LDX, li, 0, 0, h2);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const bf16x8 bn = ks + 1 < 16 ? row_frag(A0, LDX, li, 0, ks + 1, h2) : b;
            Frag8 w0;
            GS_TAKE(ws, 64 + ks, w0);
            oa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0.v, b, oa, 0, 0, 0);
            b = bn;
            __builtin_amdgcn_sched_barrier(0);
        }
        float t[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = oa[r];
        // (column block w of DO: its LN2 addends were consumed by this wave's own col_sums above)
        stage_rows(DO, LDX, li, qc, h2, t);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
        // body of attention_mfma.hip's backward kernel (MODE 2), q|k|v in QKV, dO in DO, everything of head `wave`
        const uint32_t qm = (uint32_t)(((1ull << Smax) - 1ull) << my_start);
        const uint32_t km = (kmask_raw << my_start) & qm;
        const uint64_t hbase = ((uint64_t)my_seq * GH + wave) * Smax - my_start;
        floatx16 acc, acc2;
        float p[16], g[16];
        // pass A: lane = (query li, half h2), registers over keys
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
#pragma unroll
        for (int step = 0; step < 2; ++step) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(QKV, LDQ, li, kc, step, h2), row_frag(QKV, LDQ, li, qc, step, h2),
                                                          acc, 0, 0, 0);       // K Q^T
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(QKV, LDQ, li, vc, step, h2), row_frag(DO, LDX, li, qc, step, h2),
                                                           acc2, 0, 0, 0);     // V dO^T
        }
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = ((km >> rowmap(r, h2)) & 1u) ? acc[r] * a.scale : -INFINITY;
            mx = fmaxf(mx, p[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = (p[r] == -INFINITY) ? 0.f : __expf(p[r] - mx);
            l += p[r];
        }
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.f / l;
        const float lse = mx + __logf(l);
        float D = 0.f;
        const uint32_t hrow = attn_drop_row(dp, hbase + li, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] *= inv;                                                                    // P[q][key]
            g[r] = acc2[r] * attn_drop_key(dp, hrow, (uint32_t)(rowmap(r, h2) - my_start));  // dP[q][key]
            D = fmaf(p[r], g[r], D);
        }
        D += __shfl_xor(D, 32, 64);
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = live ? p[r] * (g[r] - D) * a.scale : 0.f;       // scale * dS[q][key]
        if (h2 == 0) { my_stat[li * 3 + 0] = lse; my_stat[li * 3 + 1] = D; my_stat[li * 3 + 2] = __uint_as_float(hrow); }
        floatx16 dq;
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(QKV, LDQ, kc, ks, lane), pack_regs(g, ks), dq, 0, 0, 0);
        uint32_t dq_pk[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) dq_pk[c] = f2bf_pk(dq[2 * c], dq[2 * c + 1]);
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (my_stat written above is read below by other lanes of this wave)

        // pass B: lane = (key li, half h2), registers over queries
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
#pragma unroll
        for (int step = 0; step < 2; ++step) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(QKV, LDQ, li, qc, step, h2), row_frag(QKV, LDQ, li, kc, step, h2),
                                                          acc, 0, 0, 0);       // Q K^T
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(DO, LDX, li, qc, step, h2), row_frag(QKV, LDQ, li, vc, step, h2),
                                                           acc2, 0, 0, 0);     // dO V^T
        }
        const bool kvalid = live && (bool)((km >> li) & 1u);
        uint32_t pp[8], gp[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float pv[2], gv[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = 2 * c + e;
                const int q = rowmap(r, h2);
                const float lse_q = my_stat[q * 3 + 0], D_q = my_stat[q * 3 + 1];
                const bool ok = kvalid && q < S && ((qm >> q) & 1u);
                const float pr = ok ? __expf(acc[r] * a.scale - lse_q) : 0.f;                // P[q][key = li]
                const float mult = attn_drop_key(dp, __float_as_uint(my_stat[q * 3 + 2]), (uint32_t)(li - my_start));
                pv[e] = pr * mult;                                                           // P~ (as used by O = P~ V)
                gv[e] = ok ? pr * (acc2[r] * mult - D_q) * a.scale : 0.f;                    // scale * dS[q][key]
            }
            pp[c] = f2bf_pk(pv[0], pv[1]);
            gp[c] = f2bf_pk(gv[0], gv[1]);
        }
        floatx16 dk, dv;
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Frag8 fg, fp;
            fg.u = make_uint4(gp[4 * ks], gp[4 * ks + 1], gp[4 * ks + 2], gp[4 * ks + 3]);
            fp.u = make_uint4(pp[4 * ks], pp[4 * ks + 1], pp[4 * ks + 2], pp[4 * ks + 3]);
            dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(QKV, LDQ, qc, ks, lane), fg.v, dk, 0, 0, 0);
            dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(DO, LDX, qc, ks, lane), fp.v, dv, 0, 0, 0);
        }
        // every operand read of this head's slabs is done (same wave, in-order LDS): stage dq | dk | dv over q | k | v
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int c = 0; c < 4; ++c)
            *reinterpret_cast<uint2*>(&QKV[li * LDQ + qc + 8 * c + 4 * h2]) = make_uint2(dq_pk[2 * c], dq_pk[2 * c + 1]);
        float t[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = dk[r];
        stage_rows(QKV, LDQ, li, kc, h2, t);
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = dv[r];
        stage_rows(QKV, LDQ, li, vc, h2, t);
    }
    lds_barrier();                                       // B3: dq | dk | dv of all heads
    store_image(a.dqkv + row0 * (3 * GD), 3 * GD, QKV, LDQ, 0, S, 3 * GD);
    if (warm_next && (a.warm & 2)) gs_warm_l2(warm_next, wave, lane, dump, 1);

    // ---- phase 4: dxn1 = dqkv . Win (column block w), LayerNorm 1 backward, + dx1 -> dx ---------------------------------------------
    {
        floatx16 ga;
#pragma unroll
        for (int r = 0; r < 16; ++r) ga[r] = 0.f;
        bf16x8 b = row_frag(QKV, LDQ, li, 0, 0, h2);
#pragma unroll
        for (int ks = 0; ks < 48; ++ks) {
            const bf16x8 bn = ks + 1 < 48 ? row_frag(QKV, LDQ, li, 0, ks + 1, h2) : b;
            Frag8 w0;
            GS_TAKE(ws, 80 + ks, w0);
            ga = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0.v, b, ga, 0, 0, 0);
            b = bn;
            __builtin_amdgcn_sched_barrier(0);
        }
        float g[16], xv[16], t1[16], t2[16], o[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = bf2f(f2bf(ga[r]));
#pragma unroll
        for (int c = 0; c < 4; ++c) unpack4(xq[c], &xv[4 * c]);
        ln_bwd_rows(g, xv, mean1, rstd1, sg1, qc, h2, dx1v, stat, wave, li, live, o, t1, t2);      // (barrier B4a inside)
        // every wave is past B4a: A0 (dao product), A1 / DO (col_sums, attention) and HI (dx1 stores) are free
        stage_rows(A0, LDX, li, qc, h2, t1);
        stage_rows(A1, LDX, li, qc, h2, t2);
        stage_rows(HI, LDX, li, qc, h2, o);
    }
    lds_barrier();                                       // B4b
    col_sums(A0, A1, qc, lane, part + 512);              // dgamma1 | dbeta1
    if (a.dx) store_image(a.dx + row0 * GD, GD, HI, LDX, 0, S, GD);
}

__device__ __forceinline__ GsBwdArgs gs_globalize(GsBwdArgs a) {
#define G(f) a.f = gs_global(a.f)
    G(dx2); G(img); G(x); G(mean1); G(rstd1); G(qkv); G(x1); G(mean2); G(rstd2); G(h); G(g1); G(g2); G(key_mask); G(seed);
    G(dx); G(dx1); G(dym); G(dpre); G(dx1m); G(dqkv); G(dg); G(ln_part);
#undef G
    return a;
}

template <int PF>
__global__ __launch_bounds__(512, 2) void gs_layer_bwd_kernel(const GsBwdArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    gs_bwd_layer<PF>(a, smem, true, nullptr);
}

// the backward pass of a whole stack in one launch: layer[0] is the LAST layer of the stack (the first one walked), see
// gs_stack_fwd_kernel
struct GsBwdStackArgs {
    GsBwdArgs layer[GS_STACK_MAX];
    int n_layers;
};
template <int PF>
__global__ __launch_bounds__(512, 2) void gs_stack_bwd_kernel(const GsBwdStackArgs A) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int n = A.n_layers;
#pragma unroll 1
    for (int l = 0; l < n; ++l) {
        const GsBwdArgs a = gs_globalize(gs_karg<GsBwdArgs>(l));
        const bf16_t* next = l + 1 < n ? gs_karg_ptr((size_t)(l + 1) * sizeof(GsBwdArgs) + offsetof(GsBwdArgs, img)) : nullptr;
        gs_bwd_layer<PF>(a, smem, l == 0, next);
        // the layer's last reads of A0 / A1 (col_sums) and HI (the dx store) sit behind its barrier B4b; the next layer's phase 0 overwrites A0 / A1
        lds_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The latent chain between the two group stages (round 4): the ResNet on the pooled encoder output and the bottleneck linear,
//     z_i = z_{i-1} + relu(W_i z_{i-1} + b_i), i = 1 .. n_res;     out = W_b z_{n_res} + b_b
// (deepsvg/model/basic_blocks.py:59-65 ResNet, deepsvg/model/model.py:193-198 Bottleneck; N = one row per icon).  As separate
// launches this is 2 per block forward (GEMM, add) and 3 per block backward (gate, input-gradient GEMM, weight-gradient GEMM)
// on 512 rows: 9 + 15 launches of 5-15 us, ~0.25 ms of a 6.4 ms step for 0.2 % of its FLOPs.  Here: ONE forward and ONE
// backward launch (+ one grouped launch for the five weight gradients), a workgroup per 32 rows, its 8 waves split the 256
// output features; the running row tile never leaves LDS.
//   forward   A operands = weight fragments read straight from the row-major bf16 weight image (a lane's 8 consecutive k of
//             one output feature are 16 contiguous bytes: no packing pass), a whole block ahead; B operands = the tile's rows.
//   backward  dz = dpre . W needs W TRANSPOSED fragments: the weight matrix passes through LDS in slabs of 32 rows (coalesced
//             16-byte loads, a whole matrix ahead in registers) and comes back through hardware-transposed reads (col_frag); the K
//             order of those reads is rowmap-permuted, so the B operand takes the tile's row pieces in the same order.
// Every intermediate is rounded to bf16 where the unfused launches store bf16 (r_i, z_i, dpre_i, dz_i): same values.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int LC_MAX = 5;              // matrices of a chain: up to 4 residual blocks + the final linear
struct LatentFwdArgs {
    const bf16_t* z0; const bf16_t* w[LC_MAX]; const float* b[LC_MAX];
    bf16_t* z[LC_MAX - 1]; bf16_t* r[LC_MAX - 1];       // training: z_1 .. z_n and the ReLU outputs r_1 .. r_n (else null)
    bf16_t* out;
    int n_rows, n_res;
};
struct LatentBwdArgs {
    const bf16_t* dout; const bf16_t* w[LC_MAX]; const bf16_t* r[LC_MAX - 1];
    bf16_t* dpre[LC_MAX - 1];           // dpre_i = dz_i where r_i > 0 (the token-major operand of dW_i)
    bf16_t* dz0;
    int n_rows, n_res;
};
// entry i of a small pointer array that lives in the kernel arguments (a run-time index into the by-value struct would move the
// array to scratch memory: a chain of scalar selects instead)
template <typename T>
__device__ __forceinline__ T lc_pick(T const (&v)[LC_MAX], int i) {
    return i == 0 ? v[0] : i == 1 ? v[1] : i == 2 ? v[2] : i == 3 ? v[3] : v[4];
}
template <typename T>
__device__ __forceinline__ T lc_pick4(T const (&v)[LC_MAX - 1], int i) {
    return i == 0 ? v[0] : i == 1 ? v[1] : i == 2 ? v[2] : v[3];
}
typedef uint32_t lc_u32x4 __attribute__((ext_vector_type(4)));      // (a native vector: arrays of HIP's uint4 STRUCT are copied by
                                                                    // memcpy, which keeps register arrays in scratch memory)
constexpr int LC_IMG = 32 * LDX * 2;                    // bytes of one [32][LDX] image
constexpr int LC_FWD_LDS = 4 * LC_IMG + LC_MAX * GD * 4;            // Z[2] | R[2] | biases
constexpr int LC_BWD_LDS = 5 * LC_IMG;                              // G[2] | DP | WS[2]

// Round 6: the chain runs on 16 workgroups (2 per XCD) and walks 5 matrices (640 KiB) that are cold in L2 - one dependent HBM
// round trip after the other through two CUs per XCD.  Workgroups past the row tiles are PREFETCHERS: 8 per matrix (block b
// runs on XCD b % 8, so 8 consecutive blocks cover the 8 L2s), each reads its matrix once and exits; the chain's own loads
// then find the lines in L2.
__device__ __forceinline__ void lc_prefetch_matrix(const bf16_t* w) {
    const uint4* p = reinterpret_cast<const uint4*>(w) + threadIdx.x;
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < GD * GD * 2 / (512 * 16); ++i) {
        const uint4 v = p[i * 512];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    asm volatile("" ::"v"(acc));
}

__global__ __launch_bounds__(512, 2) void latent_chain_fwd_kernel(const LatentFwdArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    {
        const int n_tiles = (a.n_rows + 31) / 32;
        if ((int)blockIdx.x >= n_tiles) {
            lc_prefetch_matrix(a.w[((int)blockIdx.x - n_tiles) >> 3]);
            return;
        }
    }
    auto Zi = [&](int i) -> bf16_t* { return reinterpret_cast<bf16_t*>(smem + i * LC_IMG); };           // Z[0], Z[1]
    auto Ri = [&](int i) -> bf16_t* { return reinterpret_cast<bf16_t*>(smem + (2 + i) * LC_IMG); };     // R[0], R[1]
    float* sb = reinterpret_cast<float*>(smem + 4 * LC_IMG);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, h2 = lane >> 5;
    const long long row0 = (long long)blockIdx.x * 32;
    const int S = (int)min((long long)32, a.n_rows - row0);
    const int oc = wave * 32;                            // this wave's output-feature block
    const bool train = a.z[0] != nullptr;
    // the weights do not depend on the activations: the 16 fragments of block blk + 1 are fetched while block blk computes (the
    // launch is 16 workgroups deep in cold-memory latency, ~2.5 us per dependent round trip: one per block instead of three)
    auto wfetch = [&](int blk, uint4 (&f)[16]) __attribute__((always_inline)) {
        const char* wrow = reinterpret_cast<const char*>(lc_pick(a.w, blk)) + ((size_t)(oc + li) * GD + 8 * h2) * 2;
#pragma unroll
        for (int i = 0; i < 16; ++i) f[i] = *reinterpret_cast<const uint4*>(wrow + 32 * i);
    };
    uint4 wcur[16], wnext[16];
    wfetch(0, wnext);
    load_image(Zi(0), LDX, 0, a.z0 + row0 * GD, GD, S, GD);
    for (int i = 0; i <= a.n_res; ++i)
        if (tid < GD) sb[i * GD + tid] = lc_pick(a.b, i)[tid];
    lds_barrier();
    int cur = 0;
    for (int blk = 0; blk <= a.n_res; ++blk) {
#pragma unroll
        for (int i = 0; i < 16; ++i) wcur[i] = wnext[i];
        if (blk < a.n_res) wfetch(blk + 1, wnext);
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            Frag8 wf;
            wf.u = wcur[ks];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf.v, row_frag(Zi(cur), LDX, li, 0, ks, h2), acc, 0, 0, 0);
        }
        const float* bb = sb + blk * GD + oc;
        float t[16];
        if (blk < a.n_res) {
            float zv[16], rv[16];
            load_rows(Zi(cur), LDX, li, oc, h2, zv);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                rv[r] = bf2f(f2bf(fmaxf(acc[r] + bb[rowmap(r, h2)], 0.f)));     // (the unfused GEMM stores r as bf16)
                t[r] = zv[r] + rv[r];
            }
            stage_rows(Zi(cur ^ 1), LDX, li, oc, h2, t);
            if (train) stage_rows(Ri(blk & 1), LDX, li, oc, h2, rv);
            lds_barrier();
            if (train) {
                store_image(lc_pick4(a.z, blk) + row0 * GD, GD, Zi(cur ^ 1), LDX, 0, S, GD);
                store_image(lc_pick4(a.r, blk) + row0 * GD, GD, Ri(blk & 1), LDX, 0, S, GD);
            }
            cur ^= 1;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = acc[r] + bb[rowmap(r, h2)];
            stage_rows(Zi(cur ^ 1), LDX, li, oc, h2, t);
            lds_barrier();
            store_image(a.out + row0 * GD, GD, Zi(cur ^ 1), LDX, 0, S, GD);
        }
    }
}

// B operand of the transposed-weight products: B[k slot e of lane half h2][n = row] = img[row][col0 + rowmap(8 ks + e, h2)]
// (the K order col_frag delivers its A operand in): the two 8-byte pieces at columns 16 ks + 4 h2 and 16 ks + 8 + 4 h2
__device__ __forceinline__ bf16x8 row_frag_perm(const bf16_t* img, int ld, int row, int col0, int ks, int h2) {
    Frag8 f;
    const uint2 lo = *reinterpret_cast<const uint2*>(&img[row * ld + col0 + 16 * ks + 4 * h2]);
    const uint2 hi = *reinterpret_cast<const uint2*>(&img[row * ld + col0 + 16 * ks + 8 + 4 * h2]);
    f.u = make_uint4(lo.x, lo.y, hi.x, hi.y);
    return f.v;
}

__global__ __launch_bounds__(512, 2) void latent_chain_bwd_kernel(const LatentBwdArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    {
        const int n_tiles = (a.n_rows + 31) / 32;
        if ((int)blockIdx.x >= n_tiles) {       // prefetcher workgroups (see lc_prefetch_matrix)
            lc_prefetch_matrix(a.w[((int)blockIdx.x - n_tiles) >> 3]);
            return;
        }
    }
    auto Gi = [&](int i) -> bf16_t* { return reinterpret_cast<bf16_t*>(smem + i * LC_IMG); };           // G[0], G[1]
    bf16_t* DP = reinterpret_cast<bf16_t*>(smem + 2 * LC_IMG);
    auto WSi = [&](int i) -> bf16_t* { return reinterpret_cast<bf16_t*>(smem + (3 + i) * LC_IMG); };    // WS[0], WS[1]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, h2 = lane >> 5;
    const long long row0 = (long long)blockIdx.x * 32;
    const int S = (int)min((long long)32, a.n_rows - row0);
    const int oc = wave * 32;
    const int er = tid >> 4, ec = (tid & 15) * 16;       // elementwise mapping: row er, columns ec .. ec + 15
    const long long erow = row0 + min(er, S - 1);
    // a thread's 16 elements of every 32-row slab of a matrix: 8 slabs x 2 pieces; the NEXT step's matrix is fetched while this
    // step's products run (the weights do not depend on the chain), so a step costs LDS traffic and barriers, not memory latency
    auto mfetch = [&](int mi, lc_u32x4 (&fa)[8], lc_u32x4 (&fb)[8]) __attribute__((always_inline)) {
        const bf16_t* wsrc = lc_pick(a.w, mi) + (size_t)er * GD + ec;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            fa[s] = *reinterpret_cast<const lc_u32x4*>(wsrc + (size_t)s * 32 * GD);
            fb[s] = *reinterpret_cast<const lc_u32x4*>(wsrc + (size_t)s * 32 * GD + 8);
        }
    };
    // two register sets, A and B, that swap roles from step to step (no copies: a copy loop between arrays makes the compiler keep
    // all four in scratch memory); the step loop is fully unrolled over the at most LC_MAX steps
    lc_u32x4 A0[8], A1[8], B0[8], B1[8];
    mfetch(a.n_res, A0, A1);
    load_image(DP, LDX, 0, a.dout + row0 * GD, GD, S, GD);
    int cur = 0;
    auto do_step = [&](int step, lc_u32x4 (&ca)[8], lc_u32x4 (&cb)[8], lc_u32x4 (&na)[8], lc_u32x4 (&nb)[8]) __attribute__((always_inline)) {
        const int mi = a.n_res - step;                  // matrix of this step: the final linear first, then the blocks backwards
        if (mi > 0) {
            const bf16_t* wsrc = lc_pick(a.w, mi - 1) + (size_t)er * GD + ec;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                na[s] = *reinterpret_cast<const lc_u32x4*>(wsrc + (size_t)s * 32 * GD);
                nb[s] = *reinterpret_cast<const lc_u32x4*>(wsrc + (size_t)s * 32 * GD + 8);
            }
        }
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // (the 8 slabs written out: a loop with a barrier inside is unrolled too late for the register arrays to be split into
        // registers - they would live in scratch memory)
#define DSVG_LC_SLAB(s)                                                                                                \
        {                                                                                                              \
            bf16_t* ws = WSi((s) & 1);                                                                                 \
            *reinterpret_cast<lc_u32x4*>(&ws[er * LDX + ec]) = ca[s];                                                  \
            *reinterpret_cast<lc_u32x4*>(&ws[er * LDX + ec + 8]) = cb[s];                                              \
            lds_barrier();      /* slab s (and, for s = 0, the DP image) complete; the buffer was last read two slabs ago */ \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(ws, LDX, oc, 0, lane),                               \
                                                         row_frag_perm(DP, LDX, li, 32 * (s), 0, h2), acc, 0, 0, 0);     \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(ws, LDX, oc, 1, lane),                               \
                                                         row_frag_perm(DP, LDX, li, 32 * (s), 1, h2), acc, 0, 0, 0);     \
        }
        DSVG_LC_SLAB(0) DSVG_LC_SLAB(1) DSVG_LC_SLAB(2) DSVG_LC_SLAB(3) DSVG_LC_SLAB(4) DSVG_LC_SLAB(5) DSVG_LC_SLAB(6) DSVG_LC_SLAB(7)
#undef DSVG_LC_SLAB
        // acc[r] = (DP . W)[row li][column oc + rowmap(r, h2)]
        float t[16];
        if (step == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = acc[r];
        } else {
            float gv[16];
            load_rows(Gi(cur), LDX, li, oc, h2, gv);
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = gv[r] + acc[r];
        }
        stage_rows(Gi(cur ^ 1), LDX, li, oc, h2, t);
        cur ^= 1;
        lds_barrier();              // dz of this step complete; every wave is done with DP
        if (mi > 0) {
            // dpre_{mi} = dz where r_{mi} > 0: elementwise, into DP (the next product's B operand) and to memory
            const bf16_t* rr = lc_pick4(a.r, mi - 1) + erow * GD + ec;
            const uint4 ra = *reinterpret_cast<const uint4*>(rr), rb = *reinterpret_cast<const uint4*>(rr + 8);
            const uint4 ga = *reinterpret_cast<const uint4*>(&Gi(cur)[er * LDX + ec]);
            const uint4 gb = *reinterpret_cast<const uint4*>(&Gi(cur)[er * LDX + ec + 8]);
            auto gate = [](uint4 g, uint4 r) -> uint4 {
                const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, rw[4] = {r.x, r.y, r.z, r.w};
                uint32_t o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // bf16 halves: keep where r > 0 (sign clear and non-zero)
                    const uint32_t lo_on = ((rw[e] & 0x8000u) == 0u && (rw[e] & 0x7fffu) != 0u) ? 0x0000ffffu : 0u;
                    const uint32_t hi_on = ((rw[e] & 0x80000000u) == 0u && (rw[e] & 0x7fff0000u) != 0u) ? 0xffff0000u : 0u;
                    o[e] = gw[e] & (lo_on | hi_on);
                }
                return make_uint4(o[0], o[1], o[2], o[3]);
            };
            const uint4 pa = gate(ga, ra), pb = gate(gb, rb);
            *reinterpret_cast<uint4*>(&DP[er * LDX + ec]) = pa;
            *reinterpret_cast<uint4*>(&DP[er * LDX + ec + 8]) = pb;
            if (er < S) {
                bf16_t* dd = lc_pick4(a.dpre, mi - 1) + (row0 + er) * GD + ec;
                *reinterpret_cast<uint4*>(dd) = pa;
                *reinterpret_cast<uint4*>(dd + 8) = pb;
            }
            // (the next step's first slab barrier orders these DP writes in front of its reads)
        } else {
            store_image(a.dz0 + row0 * GD, GD, Gi(cur), LDX, 0, S, GD);
        }
    };
    // (written out: a loop over `step`, even a fully unrolled one, selects the arrays through pointers and they stay in scratch)
    static_assert(LC_MAX == 5, "the step sequence below is written out for at most 5 matrices");
    do_step(0, A0, A1, B0, B1);
    if (a.n_res >= 1) {
        do_step(1, B0, B1, A0, A1);
        if (a.n_res >= 2) {
            do_step(2, A0, A1, B0, B1);
            if (a.n_res >= 3) {
                do_step(3, B0, B1, A0, A1);
                if (a.n_res >= 4) do_step(4, A0, A1, B0, B1);
            }
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int64_t dsvg_gs_pack_bytes(int32_t n_layers) { return (int64_t)n_layers * GH * GS_FRAGS * FRAG; }

extern "C" int dsvg_gs_pack(const float* flat_f32, const int64_t* offs, int32_t n_layers, int32_t d_model, int32_t d_ff,
                            int32_t n_heads, void* packed_fwd, void* packed_bwd, void* stream) {
    DSVG_CHECK_ARG(flat_f32 && offs && packed_fwd && packed_bwd, "gs_pack: null pointer");
    DSVG_CHECK_ARG(d_model == GD && d_ff == GF && n_heads == GH,
                   "gs_pack: the fused group-stage kernels are built for d_model 256 / dim_ff 512 / 8 heads");
    DSVG_CHECK_ARG(n_layers > 0, "gs_pack: bad layer count");
    const long long n = (long long)n_layers * 2 * GH * GS_FRAGS * 64;
    hipLaunchKernelGGL(gs_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flat_f32, offs,
                       n_layers, (bf16_t*)packed_fwd, (bf16_t*)packed_bwd);
    DSVG_LAUNCH_CHECK("gs_pack");
    return 0;
}

// Whole sequences per 32-row tile.  32 / S fills the MFMA tiles, but a 4096-row stage is then 128 workgroups on 256 CUs, each
// bound by streaming the layer's 1-1.5 MiB of weights through ONE CU (~50 GB/s of the ~110 a CU ingests; the matrix pipe is busy
// 5 % of the time).  Round 6: while the launch would leave half of the CUs idle, a tile takes half the sequences - the other
// MFMA rows idle, twice the CUs stream.  Results per row do not depend on the tiling (dropout draws are indexed by row); the
// LayerNorm parameter gradients are sums of per-tile partials in tile order (another grouping of the same terms).
// DSVG_GS_FILL=0: always 32 / S (A/B knob).
static int gs_per(int64_t n_seq, int S) {
    static const int fill = getenv("DSVG_GS_FILL") ? atoi(getenv("DSVG_GS_FILL")) : 0;
    int per = 32 / S;
    while (fill && per >= 2 && (n_seq + per - 1) / per <= 128) per /= 2;
    return per;
}

static int gs_warm() {
    static const int w = getenv("DSVG_GS_WARM") ? atoi(getenv("DSVG_GS_WARM")) : 1;
    return w;
}

static unsigned long long* g_gs_dbg_host = nullptr;
/* development probe: buf = device buffer of (workgroups * 8 waves * 8) uint64 or NULL (off); see gs_stamp */
extern "C" int dsvg_gs_debug_clock(void* buf) {
    g_gs_dbg_host = (unsigned long long*)buf;
    return 0;
}

extern "C" int dsvg_gs_layer_fwd(const void* x, const void* packed_fwd_layer, const float* in_bias, const float* out_bias,
                                 const float* b1, const float* b2, const float* gamma1, const float* beta1,
                                 const float* gamma2, const float* beta2, const uint64_t* key_mask, const void* seq_add,
                                 int64_t seq_add_ld, int64_t n_seq, int32_t S, void* x2, float* mean1, float* rstd1, void* xn1, void* qkv,
                                 void* ao, void* x1, float* mean2, float* rstd2, void* xn2, void* h, float eps,
                                 float scale, float drop_p, uint32_t site0, const void* seed, int64_t seq_base,
                                 int32_t ffn_format, void* stream) {
    DSVG_CHECK_ARG(x && packed_fwd_layer && in_bias && out_bias && b1 && b2 && gamma1 && beta1 && gamma2 && beta2 && x2,
                   "gs_layer_fwd: null pointer");
    DSVG_CHECK_ARG(S >= 1 && S <= 32, "gs_layer_fwd: sequences of 1 .. 32 rows (got %d)", S);
    DSVG_CHECK_ARG(seq_base >= 0 && (seq_base + n_seq) * S < (1ll << 31), "gs_layer_fwd: bad sequence offset");
    DSVG_CHECK_ARG(n_seq > 0 && n_seq * S < (1ll << 31), "gs_layer_fwd: bad sizes");
    DSVG_CHECK_ARG(!(drop_p > 0.f) || seed, "gs_layer_fwd: dropout needs a seed");
    DSVG_CHECK_ARG(!seq_add || (seq_add_ld >= GD && seq_add_ld % 8 == 0), "gs_layer_fwd: bad seq_add row stride %lld", (long long)seq_add_ld);
    const bool train = xn1 != nullptr;
    DSVG_CHECK_ARG(!train || (mean1 && rstd1 && qkv && ao && x1 && mean2 && rstd2 && xn2 && h),
                   "gs_layer_fwd: the training outputs come together");
    DSVG_CHECK_ARG((((uintptr_t)x | (uintptr_t)x2 | (uintptr_t)packed_fwd_layer | (uintptr_t)xn1 | (uintptr_t)qkv |
                     (uintptr_t)ao | (uintptr_t)x1 | (uintptr_t)xn2 | (uintptr_t)h | (uintptr_t)seq_add) & 15) == 0,
                   "gs_layer_fwd: operands must be 16-byte aligned");
    GsFwdArgs a;
    a.x = (const bf16_t*)x; a.img = (const bf16_t*)packed_fwd_layer;
    a.in_bias = in_bias; a.out_bias = out_bias; a.b1 = b1; a.b2 = b2;
    a.g1 = gamma1; a.be1 = beta1; a.g2 = gamma2; a.be2 = beta2;
    a.key_mask = key_mask; a.gadd = (const bf16_t*)seq_add; a.seed = (const uint64_t*)seed;
    a.gadd_ld = seq_add ? (long long)seq_add_ld : GD;
    a.x2 = (bf16_t*)x2; a.mean1 = mean1; a.rstd1 = rstd1; a.xn1 = (bf16_t*)xn1; a.qkv = (bf16_t*)qkv; a.ao = (bf16_t*)ao;
    a.x1 = (bf16_t*)x1; a.mean2 = mean2; a.rstd2 = rstd2; a.xn2 = (bf16_t*)xn2; a.h = (bf16_t*)h;
    a.n_seq = (int)n_seq; a.S = S; a.eps = eps; a.scale = scale; a.drop_p = drop_p; a.site0 = site0;
    a.seq_base = seq_base; a.ffn_format = ffn_format; a.dbg = g_gs_dbg_host;
    const int per = gs_per(n_seq, S);
    a.per = per;
    a.warm = gs_warm();
    const int nb = (int)((n_seq + per - 1) / per);
    hipStream_t st = (hipStream_t)stream;
    static const int pf = getenv("DSVG_GS_PF") ? atoi(getenv("DSVG_GS_PF")) : GS_PF_DEFAULT;       // tuning knob
#define DSVG_GS_FWD(TR, P)                                                                       \
    do {                                                                                         \
        DSVG_ENSURE_LDS((gs_layer_fwd_kernel<TR, P>), F_LDS + GS_WARM_LDS);                                    \
        hipLaunchKernelGGL((gs_layer_fwd_kernel<TR, P>), dim3(nb), dim3(512), F_LDS + GS_WARM_LDS, st, a);     \
    } while (0)
    if (train) { if (pf == 8) DSVG_GS_FWD(true, 8); else if (pf == 24) DSVG_GS_FWD(true, 24); else if (pf == 16) DSVG_GS_FWD(true, 16); else DSVG_GS_FWD(true, 12); }
    else { if (pf == 8) DSVG_GS_FWD(false, 8); else if (pf == 24) DSVG_GS_FWD(false, 24); else if (pf == 16) DSVG_GS_FWD(false, 16); else DSVG_GS_FWD(false, 12); }
#undef DSVG_GS_FWD
    DSVG_LAUNCH_CHECK("gs_layer_fwd");
    return 0;
}

extern "C" int64_t dsvg_gs_bwd_workspace_bytes(int64_t n_seq, int32_t S) {
    const int per = gs_per(n_seq, S > 0 ? S : 1);
    return ((n_seq + per - 1) / per) * 1024 * (int64_t)sizeof(float);
}

extern "C" int dsvg_gs_layer_bwd(const void* dx2, const void* packed_bwd_layer, const void* x, const float* mean1,
                                 const float* rstd1, const void* qkv, const void* x1, const float* mean2,
                                 const float* rstd2, const void* h, const float* gamma1, const float* gamma2,
                                 const uint64_t* key_mask, int64_t n_seq, int32_t S, void* dx, void* dx1, void* dym,
                                 void* dpre, void* dx1m, void* dqkv, float* dgamma2, float* dbeta2, float* dgamma1,
                                 float* dbeta1, float scale, float drop_p, uint32_t site0, const void* seed,
                                 void* workspace, int64_t workspace_bytes, void* dg, void* stream) {
    DSVG_CHECK_ARG(dx2 && packed_bwd_layer && x && mean1 && rstd1 && qkv && x1 && mean2 && rstd2 && h && gamma1 && gamma2,
                   "gs_layer_bwd: null input pointer");
    DSVG_CHECK_ARG(dx && dym && dpre && dx1m && dqkv && dgamma2 && dbeta2 && dgamma1 && dbeta1 && workspace,
                   "gs_layer_bwd: null output pointer");
    DSVG_CHECK_ARG(S >= 1 && S <= 32 && (32 % S) == 0, "gs_layer_bwd: sequence length must divide 32 (got %d)", S);
    DSVG_CHECK_ARG(n_seq > 0 && n_seq * S < (1ll << 31), "gs_layer_bwd: bad sizes");
    DSVG_CHECK_ARG(!(drop_p > 0.f) || seed, "gs_layer_bwd: dropout needs a seed");
    DSVG_CHECK_ARG(workspace_bytes >= dsvg_gs_bwd_workspace_bytes(n_seq, S), "gs_layer_bwd: workspace too small");
    DSVG_CHECK_ARG((((uintptr_t)dx2 | (uintptr_t)x | (uintptr_t)qkv | (uintptr_t)x1 | (uintptr_t)h | (uintptr_t)dx |
                     (uintptr_t)dx1 | (uintptr_t)dym | (uintptr_t)dpre | (uintptr_t)dx1m | (uintptr_t)dqkv |
                     (uintptr_t)packed_bwd_layer | (uintptr_t)workspace | (uintptr_t)dg) & 15) == 0,
                   "gs_layer_bwd: operands must be 16-byte aligned");
    GsBwdArgs a;
    a.dx2 = (const bf16_t*)dx2; a.img = (const bf16_t*)packed_bwd_layer;
    a.x = (const bf16_t*)x; a.mean1 = mean1; a.rstd1 = rstd1; a.qkv = (const bf16_t*)qkv;
    a.x1 = (const bf16_t*)x1; a.mean2 = mean2; a.rstd2 = rstd2; a.h = (const bf16_t*)h;
    a.g1 = gamma1; a.g2 = gamma2; a.key_mask = key_mask; a.seed = (const uint64_t*)seed;
    a.dx = (bf16_t*)dx; a.dx1 = (bf16_t*)dx1; a.dym = (bf16_t*)dym; a.dpre = (bf16_t*)dpre; a.dx1m = (bf16_t*)dx1m;
    a.dqkv = (bf16_t*)dqkv; a.dg = (bf16_t*)dg; a.dg_ld = GD; a.ln_part = (float*)workspace;
    a.n_seq = (int)n_seq; a.S = S; a.scale = scale; a.drop_p = drop_p; a.site0 = site0;
    const int per = gs_per(n_seq, S);
    a.per = per;
    a.warm = gs_warm();
    const int nb = (int)((n_seq + per - 1) / per);
    hipStream_t st = (hipStream_t)stream;
    static const int pf = getenv("DSVG_GS_PF") ? atoi(getenv("DSVG_GS_PF")) : GS_PF_DEFAULT;       // tuning knob
#define DSVG_GS_BWD(P)                                                                       \
    do {                                                                                     \
        DSVG_ENSURE_LDS((gs_layer_bwd_kernel<P>), B_LDS + GS_WARM_LDS);                                    \
        hipLaunchKernelGGL((gs_layer_bwd_kernel<P>), dim3(nb), dim3(512), B_LDS + GS_WARM_LDS, st, a);     \
    } while (0)
    if (pf == 8) DSVG_GS_BWD(8); else if (pf == 16) DSVG_GS_BWD(16); else DSVG_GS_BWD(12);
#undef DSVG_GS_BWD
    DSVG_LAUNCH_CHECK("gs_layer_bwd");
    // the four LayerNorm parameter gradients: fixed-order sums of the per-tile partials (queued while a deferral scope is
    // open on this stream, like every other parameter-gradient reduction)
    float* outs[4] = {dgamma2, dbeta2, dgamma1, dbeta1};
    for (int k = 0; k < 4; ++k) {
        const int rc = dsvg_reduce_partials_strided((const float*)workspace + 256 * k, nb, 1024, 256, outs[k], 0, st);
        if (rc) return rc;
    }
    return 0;
}

/* ---- one launch per STACK (round 6) ----------------------------------------------------------------------------------- */
extern "C" int dsvg_gs_stack_fwd(const void* x, const dsvg_gs_fwd_layer* layers, int32_t n_layers, const uint64_t* key_mask,
                                 int64_t seq_add_ld, int64_t n_seq, int32_t S, float eps, float scale, float drop_p,
                                 const void* seed, void* stream) {
    DSVG_CHECK_ARG(x && layers, "gs_stack_fwd: null pointer");
    DSVG_CHECK_ARG(n_layers >= 1 && n_layers <= GS_STACK_MAX, "gs_stack_fwd: 1 .. %d layers per launch (got %d)", GS_STACK_MAX, n_layers);
    DSVG_CHECK_ARG(S >= 1 && S <= 32, "gs_stack_fwd: sequences of 1 .. 32 rows (got %d)", S);
    DSVG_CHECK_ARG(n_seq > 0 && n_seq * S < (1ll << 31), "gs_stack_fwd: bad sizes");
    DSVG_CHECK_ARG(!(drop_p > 0.f) || seed, "gs_stack_fwd: dropout needs a seed");
    const bool train = layers[0].xn1 != nullptr;
    const int per = gs_per(n_seq, S);
    GsFwdStackArgs A;
    memset(&A, 0, sizeof(A));
    A.n_layers = n_layers;
    for (int l = 0; l < n_layers; ++l) {
        const dsvg_gs_fwd_layer& L = layers[l];
        DSVG_CHECK_ARG(L.packed_fwd_layer && L.in_bias && L.out_bias && L.b1 && L.b2 && L.gamma1 && L.beta1 && L.gamma2 && L.beta2,
                       "gs_stack_fwd: null pointer in layer %d", l);
        DSVG_CHECK_ARG((L.xn1 != nullptr) == train, "gs_stack_fwd: training outputs for every layer or for none");
        DSVG_CHECK_ARG(!train || (L.x2 && L.mean1 && L.rstd1 && L.qkv && L.ao && L.x1 && L.mean2 && L.rstd2 && L.xn2 && L.h),
                       "gs_stack_fwd: the training outputs of layer %d come together", l);
        DSVG_CHECK_ARG(L.x2 || l + 1 < n_layers, "gs_stack_fwd: the last layer needs x2");
        DSVG_CHECK_ARG(!L.seq_add || (seq_add_ld >= GD && seq_add_ld % 8 == 0), "gs_stack_fwd: bad seq_add row stride %lld", (long long)seq_add_ld);
        DSVG_CHECK_ARG((((uintptr_t)x | (uintptr_t)L.x2 | (uintptr_t)L.packed_fwd_layer | (uintptr_t)L.xn1 | (uintptr_t)L.qkv |
                         (uintptr_t)L.ao | (uintptr_t)L.x1 | (uintptr_t)L.xn2 | (uintptr_t)L.h | (uintptr_t)L.seq_add) & 15) == 0,
                       "gs_stack_fwd: operands must be 16-byte aligned");
        GsFwdArgs& a = A.layer[l];
        a.x = (const bf16_t*)(l == 0 ? x : layers[l - 1].x2);       // (read by layer 0 only)
        a.img = (const bf16_t*)L.packed_fwd_layer;
        a.in_bias = L.in_bias; a.out_bias = L.out_bias; a.b1 = L.b1; a.b2 = L.b2;
        a.g1 = L.gamma1; a.be1 = L.beta1; a.g2 = L.gamma2; a.be2 = L.beta2;
        a.key_mask = key_mask; a.gadd = (const bf16_t*)L.seq_add; a.seed = (const uint64_t*)seed;
        a.gadd_ld = L.seq_add ? (long long)seq_add_ld : GD;
        a.x2 = (bf16_t*)L.x2; a.mean1 = L.mean1; a.rstd1 = L.rstd1; a.xn1 = (bf16_t*)L.xn1; a.qkv = (bf16_t*)L.qkv;
        a.ao = (bf16_t*)L.ao; a.x1 = (bf16_t*)L.x1; a.mean2 = L.mean2; a.rstd2 = L.rstd2; a.xn2 = (bf16_t*)L.xn2; a.h = (bf16_t*)L.h;
        a.n_seq = (int)n_seq; a.S = S; a.per = per; a.warm = gs_warm();
        a.eps = eps; a.scale = scale; a.drop_p = drop_p; a.site0 = L.site0;
        a.seq_base = 0; a.ffn_format = 0;
        // development probe (dsvg_gs_debug_clock): the stamps of ONE layer of the stack, DSVG_GS_DBG_LAYER (default: the last)
        static const int dbg_layer = getenv("DSVG_GS_DBG_LAYER") ? atoi(getenv("DSVG_GS_DBG_LAYER")) : -1;
        a.dbg = (l == (dbg_layer < 0 ? n_layers - 1 : dbg_layer)) ? g_gs_dbg_host : nullptr;
    }
    const int nb = (int)((n_seq + per - 1) / per);
    hipStream_t st = (hipStream_t)stream;
    if (train) {
        DSVG_ENSURE_LDS((gs_stack_fwd_kernel<true, GS_PF_DEFAULT>), F_LDS + GS_WARM_LDS);
        hipLaunchKernelGGL((gs_stack_fwd_kernel<true, GS_PF_DEFAULT>), dim3(nb), dim3(512), F_LDS + GS_WARM_LDS, st, A);
    } else {
        DSVG_ENSURE_LDS((gs_stack_fwd_kernel<false, GS_PF_DEFAULT>), F_LDS + GS_WARM_LDS);
        hipLaunchKernelGGL((gs_stack_fwd_kernel<false, GS_PF_DEFAULT>), dim3(nb), dim3(512), F_LDS + GS_WARM_LDS, st, A);
    }
    DSVG_LAUNCH_CHECK("gs_stack_fwd");
    return 0;
}

extern "C" int dsvg_gs_stack_bwd(const void* dx2, const dsvg_gs_bwd_layer* layers, int32_t n_layers, const uint64_t* key_mask,
                                 int64_t n_seq, int32_t S, float scale, float drop_p, const void* seed, int64_t workspace_bytes,
                                 int64_t dg_ld, void* stream) {
    DSVG_CHECK_ARG(dx2 && layers, "gs_stack_bwd: null pointer");
    DSVG_CHECK_ARG(n_layers >= 1 && n_layers <= GS_STACK_MAX, "gs_stack_bwd: 1 .. %d layers per launch (got %d)", GS_STACK_MAX, n_layers);
    DSVG_CHECK_ARG(S >= 1 && S <= 32 && (32 % S) == 0, "gs_stack_bwd: sequence length must divide 32 (got %d)", S);
    DSVG_CHECK_ARG(n_seq > 0 && n_seq * S < (1ll << 31), "gs_stack_bwd: bad sizes");
    DSVG_CHECK_ARG(!(drop_p > 0.f) || seed, "gs_stack_bwd: dropout needs a seed");
    DSVG_CHECK_ARG(workspace_bytes >= dsvg_gs_bwd_workspace_bytes(n_seq, S), "gs_stack_bwd: workspaces too small");
    DSVG_CHECK_ARG(dg_ld >= GD && dg_ld % 8 == 0, "gs_stack_bwd: bad dg row stride %lld", (long long)dg_ld);
    const int per = gs_per(n_seq, S);
    GsBwdStackArgs A;
    memset(&A, 0, sizeof(A));
    A.n_layers = n_layers;
    for (int l = 0; l < n_layers; ++l) {
        const dsvg_gs_bwd_layer& L = layers[l];                      // forward order
        DSVG_CHECK_ARG(L.packed_bwd_layer && L.x && L.mean1 && L.rstd1 && L.qkv && L.x1 && L.mean2 && L.rstd2 && L.h && L.gamma1 && L.gamma2,
                       "gs_stack_bwd: null input pointer in layer %d", l);
        DSVG_CHECK_ARG(L.dym && L.dpre && L.dx1m && L.dqkv && L.dgamma2 && L.dbeta2 && L.dgamma1 && L.dbeta1 && L.workspace,
                       "gs_stack_bwd: null output pointer in layer %d", l);
        DSVG_CHECK_ARG(L.dx || l > 0, "gs_stack_bwd: layer 0 needs dx");
        DSVG_CHECK_ARG((((uintptr_t)dx2 | (uintptr_t)L.x | (uintptr_t)L.qkv | (uintptr_t)L.x1 | (uintptr_t)L.h | (uintptr_t)L.dx |
                         (uintptr_t)L.dx1 | (uintptr_t)L.dym | (uintptr_t)L.dpre | (uintptr_t)L.dx1m | (uintptr_t)L.dqkv |
                         (uintptr_t)L.packed_bwd_layer | (uintptr_t)L.workspace | (uintptr_t)L.dg) & 15) == 0,
                       "gs_stack_bwd: operands must be 16-byte aligned");
        GsBwdArgs& a = A.layer[n_layers - 1 - l];                    // walked from the last layer to the first
        a.dx2 = (const bf16_t*)dx2;                                  // (read by the first layer walked only)
        a.img = (const bf16_t*)L.packed_bwd_layer;
        a.x = (const bf16_t*)L.x; a.mean1 = L.mean1; a.rstd1 = L.rstd1; a.qkv = (const bf16_t*)L.qkv;
        a.x1 = (const bf16_t*)L.x1; a.mean2 = L.mean2; a.rstd2 = L.rstd2; a.h = (const bf16_t*)L.h;
        a.g1 = L.gamma1; a.g2 = L.gamma2; a.key_mask = key_mask; a.seed = (const uint64_t*)seed;
        a.dx = (bf16_t*)L.dx; a.dx1 = (bf16_t*)L.dx1; a.dym = (bf16_t*)L.dym; a.dpre = (bf16_t*)L.dpre; a.dx1m = (bf16_t*)L.dx1m;
        a.dqkv = (bf16_t*)L.dqkv; a.dg = (bf16_t*)L.dg; a.dg_ld = dg_ld; a.ln_part = (float*)L.workspace;
        a.n_seq = (int)n_seq; a.S = S; a.per = per; a.warm = gs_warm();
        a.scale = scale; a.drop_p = drop_p; a.site0 = L.site0;
    }
    const int nb = (int)((n_seq + per - 1) / per);
    hipStream_t st = (hipStream_t)stream;
    DSVG_ENSURE_LDS((gs_stack_bwd_kernel<GS_PF_DEFAULT>), B_LDS + GS_WARM_LDS);
    hipLaunchKernelGGL((gs_stack_bwd_kernel<GS_PF_DEFAULT>), dim3(nb), dim3(512), B_LDS + GS_WARM_LDS, st, A);
    DSVG_LAUNCH_CHECK("gs_stack_bwd");
    // the LayerNorm parameter gradients of every layer: fixed-order sums of the per-tile partials, in the per-layer launches' order
    // (last layer first; queued while a deferral scope is open on this stream)
    for (int l = n_layers - 1; l >= 0; --l) {
        const dsvg_gs_bwd_layer& L = layers[l];
        float* outs[4] = {L.dgamma2, L.dbeta2, L.dgamma1, L.dbeta1};
        for (int k = 0; k < 4; ++k) {
            const int rc = dsvg_reduce_partials_strided((const float*)L.workspace + 256 * k, nb, 1024, 256, outs[k], 0, st);
            if (rc) return rc;
        }
    }
    return 0;
}

// prefetcher workgroups of a latent-chain launch: 8 per matrix while the row tiles leave most of the chip idle (DSVG_LC_WARM=0: none)
static unsigned lc_prefetchers(int64_t rows, int n_mats) {
    static const int on = getenv("DSVG_LC_WARM") ? atoi(getenv("DSVG_LC_WARM")) : 0;     // (round 6: measured without effect on the step, off)
    return (on && (rows + 31) / 32 <= 64) ? 8u * (unsigned)n_mats : 0u;
}

/* the latent chain (see latent_chain_fwd_kernel): weights = n_res + 1 row-major bf16 [256, 256] matrices (residual blocks, then
 * the final linear), biases fp32 [256] each; z_out / r_out: n_res training outputs each (or NULL pointers arrays' entries) */
extern "C" int dsvg_latent_chain_fwd(const void* z0, const void* const* weights, const float* const* biases, int32_t n_res,
                                     void* const* z_out, void* const* r_out, void* out, int64_t rows, void* stream) {
    DSVG_CHECK_ARG(z0 && weights && biases && out, "latent_chain_fwd: null pointer");
    DSVG_CHECK_ARG(n_res >= 0 && n_res < LC_MAX, "latent_chain_fwd: at most %d residual blocks (got %d)", LC_MAX - 1, n_res);
    DSVG_CHECK_ARG(rows > 0 && rows < (1ll << 31), "latent_chain_fwd: bad row count");
    LatentFwdArgs a;
    memset(&a, 0, sizeof(a));
    a.z0 = (const bf16_t*)z0; a.out = (bf16_t*)out; a.n_rows = (int)rows; a.n_res = n_res;
    uintptr_t al = (uintptr_t)z0 | (uintptr_t)out;
    for (int i = 0; i <= n_res; ++i) {
        DSVG_CHECK_ARG(weights[i] && biases[i], "latent_chain_fwd: null weight / bias %d", i);
        a.w[i] = (const bf16_t*)weights[i]; a.b[i] = biases[i];
        al |= (uintptr_t)weights[i];
    }
    const bool train = n_res > 0 && z_out && r_out && z_out[0];
    for (int i = 0; i < n_res; ++i) {
        a.z[i] = train ? (bf16_t*)z_out[i] : nullptr;
        a.r[i] = train ? (bf16_t*)r_out[i] : nullptr;
        DSVG_CHECK_ARG(!train || (z_out[i] && r_out[i]), "latent_chain_fwd: the training outputs come together");
        al |= (uintptr_t)a.z[i] | (uintptr_t)a.r[i];
    }
    DSVG_CHECK_ARG((al & 15) == 0, "latent_chain_fwd: operands must be 16-byte aligned");
    DSVG_ENSURE_LDS(latent_chain_fwd_kernel, LC_FWD_LDS);
    hipLaunchKernelGGL(latent_chain_fwd_kernel, dim3((unsigned)((rows + 31) / 32) + lc_prefetchers(rows, n_res + 1)), dim3(512), LC_FWD_LDS, (hipStream_t)stream, a);
    DSVG_LAUNCH_CHECK("latent_chain_fwd");
    return 0;
}

/* backward of the chain's input path: dout = dL/dout [rows, 256]; r = the n_res ReLU outputs of the forward pass; outputs:
 * dpre_out[i] = dz_{i+1} where r_{i+1} > 0 (the token-major operand of dW_{i+1}; dW of the final linear takes dout itself) and
 * dz0 = dL/dz0 */
extern "C" int dsvg_latent_chain_bwd(const void* dout, const void* const* weights, const void* const* r, int32_t n_res,
                                     void* const* dpre_out, void* dz0, int64_t rows, void* stream) {
    DSVG_CHECK_ARG(dout && weights && dz0 && (n_res == 0 || (r && dpre_out)), "latent_chain_bwd: null pointer");
    DSVG_CHECK_ARG(n_res >= 0 && n_res < LC_MAX, "latent_chain_bwd: at most %d residual blocks (got %d)", LC_MAX - 1, n_res);
    DSVG_CHECK_ARG(rows > 0 && rows < (1ll << 31), "latent_chain_bwd: bad row count");
    LatentBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.dout = (const bf16_t*)dout; a.dz0 = (bf16_t*)dz0; a.n_rows = (int)rows; a.n_res = n_res;
    uintptr_t al = (uintptr_t)dout | (uintptr_t)dz0;
    for (int i = 0; i <= n_res; ++i) {
        DSVG_CHECK_ARG(weights[i], "latent_chain_bwd: null weight %d", i);
        a.w[i] = (const bf16_t*)weights[i];
        al |= (uintptr_t)weights[i];
    }
    for (int i = 0; i < n_res; ++i) {
        DSVG_CHECK_ARG(r[i] && dpre_out[i], "latent_chain_bwd: null r / dpre %d", i);
        a.r[i] = (const bf16_t*)r[i]; a.dpre[i] = (bf16_t*)dpre_out[i];
        al |= (uintptr_t)r[i] | (uintptr_t)dpre_out[i];
    }
    DSVG_CHECK_ARG((al & 15) == 0, "latent_chain_bwd: operands must be 16-byte aligned");
    DSVG_ENSURE_LDS(latent_chain_bwd_kernel, LC_BWD_LDS);
    hipLaunchKernelGGL(latent_chain_bwd_kernel, dim3((unsigned)((rows + 31) / 32) + lc_prefetchers(rows, n_res + 1)), dim3(512), LC_BWD_LDS, (hipStream_t)stream, a);
    DSVG_LAUNCH_CHECK("latent_chain_bwd");
    return 0;
}
