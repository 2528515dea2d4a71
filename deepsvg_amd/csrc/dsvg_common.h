// Device-side helpers shared by every gfx950 kernel of the DeepSVG hot path.
// gfx950 only: 64-lane wavefronts are hard-coded, no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>

#define DSVG_F32 0
#define DSVG_BF16 1

typedef uint16_t bf16_t;  // raw bfloat16 bits; arithmetic is always done in fp32

// ---------------------------------------------------------------------------------------------
// error plumbing for the C-ABI (thread-local message, negative return codes)
// ---------------------------------------------------------------------------------------------
extern "C" const char* dsvg_last_error(void);
void dsvg_set_error(const char* fmt, ...);

#define DSVG_CHECK_ARG(cond, ...)                \
    do {                                         \
        if (!(cond)) {                           \
            dsvg_set_error(__VA_ARGS__);         \
            return -1;                           \
        }                                        \
    } while (0)

#define DSVG_LAUNCH_CHECK(name)                                                        \
    do {                                                                               \
        hipError_t _e = hipGetLastError();                                             \
        if (_e != hipSuccess) {                                                        \
            dsvg_set_error("%s: launch failed: %s", name, hipGetErrorString(_e));      \
            return -2;                                                                 \
        }                                                                              \
    } while (0)

// raise a kernel's dynamic-LDS cap above the 64 KiB default, once per high-water mark (host-side call, kept
// out of the steady state so a captured hipGraph never sees it)
#define DSVG_ENSURE_LDS(kern, bytes)                                                                      \
    do {                                                                                                  \
        static size_t _dsvg_cap = 64 * 1024;                                                              \
        if ((size_t)(bytes) > _dsvg_cap) {                                                                \
            (void)hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                      (int)(bytes));                                                      \
            _dsvg_cap = (size_t)(bytes);                                                                  \
        }                                                                                                 \
    } while (0)

// ---------------------------------------------------------------------------------------------
// bf16 <-> fp32
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// fp32 -> bf16 with the hardware converter (v_cvt_pk_bf16_f32, round-to-nearest-even)
typedef __bf16 dsvg_bf2 __attribute__((ext_vector_type(2)));
typedef float dsvg_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) {
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
// two values -> one packed dword (lo in bits 0..15)
__device__ __forceinline__ uint32_t f2bf_pk(float lo, float hi) {
    const dsvg_f2 v = {lo, hi};
    const dsvg_bf2 b = __builtin_convertvector(v, dsvg_bf2);
    return __builtin_bit_cast(uint32_t, b);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    typedef float4 raw4;  // bit-copy of 4 consecutive elements
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    // 4 consecutive elements, pointer must be 16-byte aligned
    static __device__ __forceinline__ void ld4(const float* p, float (&v)[4]) {
        float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void st4(float* p, const float (&v)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct Elem<bf16_t> {
    typedef uint2 raw4;
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
    // 4 consecutive elements, pointer must be 8-byte aligned
    static __device__ __forceinline__ void ld4(const bf16_t* p, float (&v)[4]) {
        uint2 t = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    }
    static __device__ __forceinline__ void st4(bf16_t* p, const float (&v)[4]) {
        uint2 t;
        t.x = f2bf_pk(v[0], v[1]);
        t.y = f2bf_pk(v[2], v[3]);
        *reinterpret_cast<uint2*>(p) = t;
    }
};

// 32 consecutive elements (one attention head row) -> fp32 registers; p must be 16-byte aligned
__device__ __forceinline__ void row32_load(const float* p, float (&v)[32]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 t = reinterpret_cast<const float4*>(p)[i];
        v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
    }
}
__device__ __forceinline__ void row32_load(const bf16_t* p, float (&v)[32]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint4 t = reinterpret_cast<const uint4*>(p)[i];
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[8 * i + 2 * e] = __uint_as_float(w[e] << 16);
            v[8 * i + 2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
        }
    }
}
__device__ __forceinline__ void row32_store(float* p, const float (&v)[32]) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
        reinterpret_cast<float4*>(p)[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
}
__device__ __forceinline__ void row32_store(bf16_t* p, const float (&v)[32]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            w[e] = f2bf_pk(v[8 * i + 2 * e], v[8 * i + 2 * e + 1]);
        reinterpret_cast<uint4*>(p)[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// Counter-based dropout RNG.  The mask is a pure function of (seed, site, element id) so the backward
// pass regenerates the forward mask instead of storing it; `seed` lives in device memory so that a
// captured hipGraph sees a new value on every replay.
//   group g = id >> 3 :  h  = (hash32(lo(g) ^ s0) ^ s1) + hi(g)*C                  (once per 8 elements)
//   word  i = 0..3    :  w  = fold(h * C_i),  C_i = ((0x7feb352d * (i + 1)) ^ (0x846ca68b >> i)) | 1,  fold(p) = lo32(p) ^ hi32(p)
//                                             of the 64-bit product                   (two 16-bit draws each)
//   element slot = id & 7 uses the low (even slot) / high (odd slot) half of word slot>>1 and is dropped
//   when draw16 < round(p * 65536); survivors are scaled by 65536 / (65536 - thresh16).
// One full hash round per group, then ONE multiply-fold with a per-word multiplier per pair of draws: v_mul_lo + v_mul_hi +
// v_xor.  (Round 3's word function - h + (i + 1) * 0x9e3779b9 through ONE xorshift-multiply round, 6 instructions - left a
// 0.5 % correlation between the draws of neighbouring words: tests/test_dropout_stats.py, chi-square 160 on 2^23 pairs.  The
// multiply-fold with multipliers that do NOT form an arithmetic progression - C_i = A + i B passes the pairwise tests but
// fails the per-group drop-count distribution by a wide margin: the products h C_i are then a lattice - passes every test
// there: neighbours at 1 .. 256, sites, consecutive seeds, binomial counts per group of 8 / 16 / 32 and per row; it is half the
// instructions where i is a compile-time constant.)  Restated bit-for-bit with int64 torch ops in tests/torch_ops_ref.py.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t dsvg_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
struct DropCtx {
    uint32_t s0, s1;    // mixed seed words
    uint32_t thresh;    // 16-bit threshold: drop when draw16 < thresh
    float scale;        // 65536 / (65536 - thresh)
    bool on;
};
__device__ __forceinline__ DropCtx drop_make(float p, const uint64_t* seed_ptr, uint32_t site) {
    DropCtx c;
    c.on = (p > 0.f) && (seed_ptr != nullptr);
    if (c.on) {
        uint64_t seed = *seed_ptr;
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
__device__ __forceinline__ uint32_t drop_group(const DropCtx& c, uint64_t g) {
    const uint32_t h = dsvg_hash32((uint32_t)g ^ c.s0);
    return (h ^ c.s1) + (uint32_t)(g >> 32) * 0x9e3779b1u;
}
// word i (< 16) of a group / row hash: two 16-bit draws
__device__ __forceinline__ uint32_t drop_word(uint32_t h, uint32_t i) {
    const uint32_t c = ((0x7feb352du * (i + 1u)) ^ (0x846ca68bu >> i)) | 1u;
    // (NOT `uint64_t p = (uint64_t)h * c; lo ^ hi`: hipcc 7.0 turns that into v_mad_u64_u32, and with it the bf16 training step
    // stopped being bit-reproducible from run to run on gfx950 - losses differing in the 6th digit, gradients by 5e-4, between
    // two identical steps of one process; round 5, scripts/determinism_probe.py.  v_mul_lo_u32 + v_mul_hi_u32 is reproducible.)
    return (h * c) ^ __umulhi(h, c);
}
// multiplier (0 or scale) for element idx
__device__ __forceinline__ float drop_mult(const DropCtx& c, uint64_t idx) {
    if (!c.on) return 1.f;
    const uint32_t slot = (uint32_t)idx & 7u;
    const uint32_t w = drop_word(drop_group(c, idx >> 3), slot >> 1);
    const uint32_t draw = (slot & 1u) ? (w >> 16) : (w & 0xffffu);
    return draw < c.thresh ? 0.f : c.scale;
}
// multipliers for the 8 elements idx8 .. idx8+7 (idx8 must be a multiple of 8)
__device__ __forceinline__ void drop_mult8(const DropCtx& c, uint64_t idx8, float (&m)[8]) {
    if (!c.on) {
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = 1.f;
        return;
    }
    const uint32_t h = drop_group(c, idx8 >> 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t w = drop_word(h, i);
        m[2 * i] = (w & 0xffffu) < c.thresh ? 0.f : c.scale;
        m[2 * i + 1] = (w >> 16) < c.thresh ? 0.f : c.scale;
    }
}

// Dropout of the attention probabilities ("row draws", every attention kernel + tests/torch_ops_ref.py attn_drop_mult):
// element (row, key) with row = (sequence * heads + head) * S + query and key counted inside the sequence.  ONE counter
// hash per (row, block of 32 keys) and a one-multiply finaliser per pair of keys - a query row's 32 draws cost one hash
// + 16 multiplies instead of 3 hashes per element (the per-element ids made the draw 40-60 % of an attention launch).
__device__ __forceinline__ uint32_t attn_drop_row(const DropCtx& c, uint64_t row, uint32_t key_block) {
    const uint64_t g = row * 8ull + key_block;          // <= 8 blocks = 256 keys per row
    const uint32_t h = dsvg_hash32((uint32_t)g ^ c.s0);
    return dsvg_hash32(h + (uint32_t)(g >> 32) * 0x9e3779b1u + c.s1);
}
__device__ __forceinline__ float attn_drop_key(const DropCtx& c, uint32_t hrow, uint32_t key) {
    if (!c.on) return 1.f;
    const uint32_t w = drop_word(hrow, (key & 31u) >> 1);
    const uint32_t draw = (key & 1u) ? (w >> 16) : (w & 0xffffu);
    return draw < c.thresh ? 0.f : c.scale;
}
__device__ __forceinline__ float attn_drop_mult(const DropCtx& c, uint64_t row, uint32_t key) {
    if (!c.on) return 1.f;
    return attn_drop_key(c, attn_drop_row(c, row, key >> 5), key);
}

// Gumbel noise for categorical sampling on the device (round 5): a draw from softmax(logits / T) is arg-max_c (logit_c + T g_c),
// g_c = -log(-log(u_c)) with independent uniforms u_c - the reference's _sample_categorical (deepsvg/model/utils.py:75-79:
// torch.distributions.Categorical(logits = logits / T).sample()) without the softmax, the cumulative sum, or the dense logits.
// u comes from the dropout counter hash: element (row, col) of a [rows, n_cols] logit matrix has key = row * k4 + (col >> 2)
// (k4 = ceil(n_cols / 4)) and takes word (col & 3) of that key's hash: 23 bits -> u = (bits + 1/2) 2^-23 in (0, 1), so
// g in [-2.81, 16.7] is always finite.  ONE hash per 4 consecutive columns + one multiply-fold per element.
__device__ __forceinline__ float dsvg_gumbel_from_word(uint32_t w) {
    // 23 bits: (bits + 1/2) is exact in fp32 (24 bits would round 2^24 - 1/2 up to 2^24: u = 1, g = +inf once in 2^24 draws)
    const float u = ((float)(w >> 9) + 0.5f) * (1.f / 8388608.f);
    return -__logf(-__logf(u));
}
__device__ __forceinline__ float dsvg_gumbel(const DropCtx& c, uint64_t row, uint32_t k4, uint32_t col) {
    return dsvg_gumbel_from_word(drop_word(drop_group(c, row * k4 + (col >> 2)), col & 3u));
}

// ---------------------------------------------------------------------------------------------
// wave-level reductions (64 lanes) and width-limited group reductions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int dsvg_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// split-K workspace: K slice z = [M*N partial | M row sums (only when requested)], slices `dsvg_splitk_slice` floats
// apart (rounded up to 4 floats so that every slice starts 16-byte aligned)
__host__ __device__ inline size_t dsvg_splitk_slice(size_t M, size_t N, bool rowsums) {
    return (M * N + (rowsums ? M : 0) + 3) & ~(size_t)3;
}

// out[j] = (accumulate ? out[j] : 0) + sum_{q<P} part[q*stride + j], j < n   (deterministic order; gemm.hip)
int dsvg_reduce_partials_strided(const float* part, int64_t P, int64_t stride, int64_t n, float* out,
                                 int32_t accumulate, hipStream_t st);
// same, with the columns j < n_bf16 of every slice stored as packed bf16 from the slice's start (gemm.hip)
int dsvg_reduce_partials_mixed(const float* part, int64_t P, int64_t stride, int64_t n, int64_t n_bf16, float* out,
                               int32_t accumulate, hipStream_t st);
