// GEMM with fused prologue/epilogue for the DeepSVG hot path (gfx950).
//   C = epi( sum_k A(m,k) * B(n,k) )
// fp32 path : v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain), 128x128x32 tiles, LDS
//             k-major so every operand layout (NT / NN / TN) shares one kernel.
// bf16 path : see gemm_bf16.hip (v_mfma_f32_32x32x16_bf16).
// Replaces aten::addmm/mm for every nn.Linear on the path and their autograd matmuls
// (reference call sites are listed in include/dsvg.h).
#include "dsvg_common.h"
#include "../../include/dsvg.h"
#include "gemm_common.h"
int dsvg_gemm_group_flush(hipStream_t st);       // gemm_bf16_glds.hip
#include <map>
#include <mutex>
#include <vector>

// ---------------------------------------------------------------------------------------------
// reference kernel: one thread per output element (debug / odd-stride fallback)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void gemm_naive_kernel(dsvg_gemm_desc p, int k_begin, int k_end, float* part, float* rs) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)p.M * p.N) return;
    int m = (int)(idx / p.N), n = (int)(idx % p.N);
    const T* A = (const T*)p.A;
    const T* B = (const T*)p.B;
    DropCtx adc = drop_make(p.a_drop_p, p.seed, p.a_drop_site);
    float acc = 0.f, asum = 0.f;
    for (int k = k_begin; k < k_end; ++k) {
        float a, b;
        if (p.a_kc) {
            a = Elem<T>::ld(A + (size_t)m * p.lda + k);
            a *= drop_mult(adc, (uint64_t)m * p.a_drop_ld + k);
        } else {
            a = Elem<T>::ld(A + (size_t)k * p.lda + m);
            a *= drop_mult(adc, (uint64_t)k * p.a_drop_ld + m);
        }
        b = p.b_kc ? Elem<T>::ld(B + (size_t)n * p.ldb + k) : Elem<T>::ld(B + (size_t)k * p.ldb + n);
        acc = fmaf(a, b, acc);
        asum += a;
    }
    if (rs && n == 0) rs[m] = asum;
    if (part) {
        part[(size_t)m * p.N + n] = acc;
    } else {
        DropCtx dc = drop_make(p.drop_p, p.seed, p.drop_site);
        gemm_epilogue<T>(p, dc, m, n, acc);
    }
}

// ---------------------------------------------------------------------------------------------
// fp32 MFMA kernel
//   block 256 threads = 4 waves (2x2), block tile 128x128, wave tile 64x64 = 2x2 MFMA 32x32 tiles
//   LDS: As[k][m], Bs[k][n] (k-major): fragment reads are lane-contiguous ds_read_b32
//   operand fragments (guide §3): A: lane l holds A[i=l&31][k=l>>5], B: B[k=l>>5][j=l&31]
//   C/D: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
// ---------------------------------------------------------------------------------------------
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;

__device__ __forceinline__ float4 ld4_guard(const float* p, int nvalid) {
    if (nvalid >= 4) return *reinterpret_cast<const float4*>(p);
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nvalid > 0) r.x = p[0];
    if (nvalid > 1) r.y = p[1];
    if (nvalid > 2) r.z = p[2];
    return r;
}

template <bool AKC, bool BKC, bool RS>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(dsvg_gemm_desc p, int tiles_n, int nwg_mn,
                                                            int k_chunk, float* part, float* rs_part) {
    constexpr int LDA_S = AKC ? 129 : 132;
    constexpr int LDB_S = BKC ? 129 : 132;
    __shared__ __attribute__((aligned(16))) float As[BK * LDA_S];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDB_S];

    // XCD-aware tile mapping (bijective): neighbouring column tiles of one row tile share an XCD's L2
    const int bid = blockIdx.x;
    const int xcd = bid % 8, local = bid / 8;
    int wgid, kz;
    if (gridDim.y == 1 && (int)gridDim.x != nwg_mn) {
        // split-K as a 1-D grid (nsplit % 8 == 0): the output tiles of one K slice share an XCD (and its L2)
        wgid = local % nwg_mn;
        kz = (local / nwg_mn) * 8 + xcd;
    } else {
        const int q = nwg_mn / 8, r8 = nwg_mn % 8;
        wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + local;
        kz = blockIdx.y;
    }
    const int tile_m = wgid / tiles_n, tile_n = wgid % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int k_begin = kz * k_chunk;
    const int k_end = min(p.K, k_begin + k_chunk);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const float* A = (const float*)p.A;
    const float* B = (const float*)p.B;
    const DropCtx adc = drop_make(p.a_drop_p, p.seed, p.a_drop_site);

    float4 ra[4], rb[4];
    // Fast staging path (uniform per launch): every 4-element piece is either entirely inside the matrix or entirely
    // outside, so the 8 loads of a K step are issued unconditionally from clamped addresses and zeroed afterwards.
    // (Loads under `if (inside)` are compiled as a branch + s_waitcnt vmcnt(0) each: 8 dependent round trips per step.)
    const bool fast_a = !adc.on && !(p.K & 3) && !(k_chunk & 3) && (AKC || !(p.M & 3)) && p.M >= 4 && p.K >= 4;
    const bool fast_b = !(p.K & 3) && !(k_chunk & 3) && (BKC || !(p.N & 3)) && p.N >= 4 && p.K >= 4;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    auto load_a = [&](int k0) {
        if (fast_a) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (AKC) {
                    const int gm = m0 + (tid >> 3) + 32 * j, gk = k0 + 4 * (tid & 7);
                    const float4 v = *reinterpret_cast<const float4*>(A + (size_t)min(gm, p.M - 1) * p.lda + min(gk, p.K - 4));
                    ra[j] = (gm < p.M && gk < k_end) ? v : zero4;
                } else {
                    const int gk = k0 + (tid >> 5) + 8 * j, gm = m0 + 4 * (tid & 31);
                    const float4 v = *reinterpret_cast<const float4*>(A + (size_t)min(gk, p.K - 1) * p.lda + min(gm, p.M - 4));
                    ra[j] = (gk < k_end && gm < p.M) ? v : zero4;
                }
            }
            return;
        }
        if (AKC) {
            const int c = tid & 7, r = tid >> 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gm = m0 + r + 32 * j, gk = k0 + 4 * c;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gm < p.M && gk < k_end) {
                    v = ld4_guard(A + (size_t)gm * p.lda + gk, k_end - gk);
                    if (adc.on) {
                        const uint64_t e = (uint64_t)gm * p.a_drop_ld + gk;
                        v.x *= drop_mult(adc, e); v.y *= drop_mult(adc, e + 1);
                        v.z *= drop_mult(adc, e + 2); v.w *= drop_mult(adc, e + 3);
                    }
                }
                ra[j] = v;
            }
        } else {
            const int c = tid & 31, r = tid >> 5;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gk = k0 + r + 8 * j, gm = m0 + 4 * c;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gk < k_end && gm < p.M) {
                    v = ld4_guard(A + (size_t)gk * p.lda + gm, p.M - gm);
                    if (adc.on) {
                        const uint64_t e = (uint64_t)gk * p.a_drop_ld + gm;
                        v.x *= drop_mult(adc, e); v.y *= drop_mult(adc, e + 1);
                        v.z *= drop_mult(adc, e + 2); v.w *= drop_mult(adc, e + 3);
                    }
                }
                ra[j] = v;
            }
        }
    };
    auto load_b = [&](int k0) {
        if (fast_b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (BKC) {
                    const int gn = n0 + (tid >> 3) + 32 * j, gk = k0 + 4 * (tid & 7);
                    const float4 v = *reinterpret_cast<const float4*>(B + (size_t)min(gn, p.N - 1) * p.ldb + min(gk, p.K - 4));
                    rb[j] = (gn < p.N && gk < k_end) ? v : zero4;
                } else {
                    const int gk = k0 + (tid >> 5) + 8 * j, gn = n0 + 4 * (tid & 31);
                    const float4 v = *reinterpret_cast<const float4*>(B + (size_t)min(gk, p.K - 1) * p.ldb + min(gn, p.N - 4));
                    rb[j] = (gk < k_end && gn < p.N) ? v : zero4;
                }
            }
            return;
        }
        if (BKC) {
            const int c = tid & 7, r = tid >> 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gn = n0 + r + 32 * j, gk = k0 + 4 * c;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gn < p.N && gk < k_end) v = ld4_guard(B + (size_t)gn * p.ldb + gk, k_end - gk);
                rb[j] = v;
            }
        } else {
            const int c = tid & 31, r = tid >> 5;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gk = k0 + r + 8 * j, gn = n0 + 4 * c;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gk < k_end && gn < p.N) v = ld4_guard(B + (size_t)gk * p.ldb + gn, p.N - gn);
                rb[j] = v;
            }
        }
    };
    auto store_lds = [&]() {
        if (AKC) {
            const int c = tid & 7, r = tid >> 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = r + 32 * j;
                As[(4 * c + 0) * LDA_S + m] = ra[j].x;
                As[(4 * c + 1) * LDA_S + m] = ra[j].y;
                As[(4 * c + 2) * LDA_S + m] = ra[j].z;
                As[(4 * c + 3) * LDA_S + m] = ra[j].w;
            }
        } else {
            const int c = tid & 31, r = tid >> 5;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float4*>(&As[(r + 8 * j) * LDA_S + 4 * c]) = ra[j];
        }
        if (BKC) {
            const int c = tid & 7, r = tid >> 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = r + 32 * j;
                Bs[(4 * c + 0) * LDB_S + n] = rb[j].x;
                Bs[(4 * c + 1) * LDB_S + n] = rb[j].y;
                Bs[(4 * c + 2) * LDB_S + n] = rb[j].z;
                Bs[(4 * c + 3) * LDB_S + n] = rb[j].w;
            }
        } else {
            const int c = tid & 31, r = tid >> 5;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float4*>(&Bs[(r + 8 * j) * LDB_S + 4 * c]) = rb[j];
        }
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // RS is a compile-time variant: the two extra accumulators cost 32 VGPRs, which the plain kernels must not pay
    const bool do_rs = RS && part != nullptr && rs_part != nullptr && tile_n == 0 && wn == 0;
    floatx16 accb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;

    if (k_begin < k_end) {
        load_a(k_begin);
        load_b(k_begin);
    }
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        store_lds();
        __syncthreads();
        if (k0 + BK < k_end) {  // register prefetch of the next K tile, overlapped with the MFMAs
            load_a(k0 + BK);
            load_b(k0 + BK);
        }
        const int arow = wm * 64 + (lane & 31);
        const int bcol = wn * 64 + (lane & 31);
        const int kh = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int k = 2 * kk + kh;
            const float a0 = As[k * LDA_S + arow], a1 = As[k * LDA_S + arow + 32];
            const float b0 = Bs[k * LDB_S + bcol], b1 = Bs[k * LDB_S + bcol + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            if (RS && do_rs) {   // row sums of A (bias gradient): MFMA against an all-ones B fragment
                accb[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, 1.0f, accb[0], 0, 0, 0);
                accb[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, 1.0f, accb[1], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    if (RS && do_rs && (lane & 31) == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M) rs_part[(size_t)kz * dsvg_splitk_slice(p.M, p.N, true) + m] = accb[i][r];
            }
    }
    const DropCtx dc = drop_make(p.drop_p, p.seed, p.drop_site);
    // K-slice z of the workspace = [M*N partial | M row sums (only when requested)]
    float* my_part = part ? part + (size_t)kz * dsvg_splitk_slice(p.M, p.N, rs_part != nullptr) : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int n = n0 + wn * 64 + j * 32 + (lane & 31);
                if (m < p.M && n < p.N) {
                    if (my_part) my_part[(size_t)m * p.N + n] = acc[i][j][r];
                    else gemm_epilogue<float>(p, dc, m, n, acc[i][j][r]);
                }
            }
}

// ---------------------------------------------------------------------------------------------
// deterministic reductions
// ---------------------------------------------------------------------------------------------
// out[j] = [out[j] +] sum_q part[q*stride + j].  Block = 64 column lanes x 4 slices of the partial index;
// each slice keeps 4 independent accumulators (fixed summation tree -> bit-reproducible), the 4 slices are
// combined through LDS.  VEC = 4 (float4 lanes) when stride, n and the pointers allow it.
// n_bf16 > 0 (VEC = 4 only): columns j < n_bf16 of every slice are stored as bf16, packed from the slice's start
// (element j at byte 2 j), the rest as fp32 at their usual float index - the split-K partials of the bf16 weight-gradient
// GEMM (gemm_bf16_glds.hip) with its fp32 row sums behind them.
template <int VEC, int TX, bool BF>
__global__ __launch_bounds__(256) void reduce_partials_strided_kernel(const float* __restrict__ part, long long P,
                                                                      long long stride, long long n,
                                                                      float* __restrict__ out, int accumulate,
                                                                      long long n_bf16) {
    constexpr int TY = 256 / TX;           // slices over the partial index
    __shared__ float red[TY][TX][VEC];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const long long j = ((long long)blockIdx.x * TX + tx) * VEC;
    float acc[4][VEC];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[u][e] = 0.f;
    if (j < n) {
        const bool bf = BF && j < n_bf16;
        auto ld4 = [&](long long q) -> float4 {
            if (BF && bf) {
                const uint2 t = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(part + q * stride) + j);
                return make_float4(__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u),
                                   __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u));
            }
            return *reinterpret_cast<const float4*>(part + q * stride + j);
        };
        long long q = ty;
        for (; q + 3 * TY < P; q += 4 * TY) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (VEC == 4) {
                    const float4 v = ld4(q + TY * u);
                    acc[u][0] += v.x; acc[u][1 % VEC] += v.y; acc[u][2 % VEC] += v.z; acc[u][3 % VEC] += v.w;
                } else {
                    acc[u][0] += part[(q + TY * u) * stride + j];
                }
            }
        }
        for (; q < P; q += TY) {
            if (VEC == 4) {
                const float4 v = ld4(q);
                acc[0][0] += v.x; acc[0][1 % VEC] += v.y; acc[0][2 % VEC] += v.z; acc[0][3 % VEC] += v.w;
            } else {
                acc[0][0] += part[q * stride + j];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[ty][tx][e] = (acc[0][e] + acc[1][e]) + (acc[2][e] + acc[3][e]);
    __syncthreads();
    if (ty == 0 && j < n) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < TY; ++t) s += red[t][tx][e];
            if (accumulate) s += out[j + e];
            out[j + e] = s;
        }
    }
}

int dsvg_reduce_partials_strided(const float* part, int64_t P, int64_t stride, int64_t n, float* out,
                                 int32_t accumulate, hipStream_t st) {
    return dsvg_reduce_partials_mixed(part, P, stride, n, 0, out, accumulate, st);
}

// ---------------------------------------------------------------------------------------------
// deferred reductions: inside a dsvg_defer_scope every reduction of this file's funnel is QUEUED (its partials stay in
// the caller's workspace) and dsvg_flush_deferred performs all of them in one launch per 64 queued segments - the
// parameter-gradient reductions of a backward pass (80 split-K weight gradients, 28 LayerNorm gamma/beta pairs, ...)
// are 6 us launches of 1-2 us of memory work each.  The segment table travels in the kernel arguments, so the launch
// is hipGraph-capturable without any host staging buffer.  Summation order is fixed (bit-reproducible).
// ---------------------------------------------------------------------------------------------
namespace {
struct DeferSeg {
    const float* part; float* out; long long stride;
    int n, n_bf16, P, first_block, accumulate, wide;
};
constexpr int DEFER_MAX = 64;
struct DeferTable { DeferSeg s[DEFER_MAX]; int n_seg; };
static_assert(sizeof(DeferTable) <= 3600, "the segment table must fit the kernel-argument segment");

// One queue per STREAM (a stream belongs to one device): the reductions of a launch sequence are queued on, and flushed to,
// the stream that sequence runs on, so two models / trainers / devices in one process never see each other's entries.  The
// table itself is process-wide and mutex-protected because autograd runs backward nodes on its own device thread (same
// stream, another host thread than the one that opened the scope).
struct DeferQueue {
    int scope = 0;
    std::vector<DeferSeg> q;
};
struct DeferTableOfQueues {
    std::mutex mu;
    std::map<hipStream_t, DeferQueue> by_stream;
};
DeferTableOfQueues& defer_queues() { static DeferTableOfQueues d; return d; }

constexpr int DF_TX = 32, DF_TY = 8;       // 32 float4 column lanes x 8 groups over the partial index per block

// "wide" segments (split-K slices of a weight gradient: >= 2048 columns, <= 64 slices): a thread owns 8 consecutive columns
// and walks the slices itself, 8 loads of 16 bytes in flight - a wave reads 1 KiB runs of one slice per instruction.  The
// other segments (LayerNorm / bias partials: few columns, up to thousands of partial rows) keep the 32 x 8 layout with a
// tree over 8 groups of partial rows.  Both orders are fixed (bit-reproducible).
constexpr int DF_WIDE_COLS = 256 * 8;

__device__ __forceinline__ void defer_ld8(const float* part, long long stride, int q, int j, int n_bf16, float (&v)[8]) {
    const float* row = part + q * stride;
    if (j + 8 <= n_bf16) {
        const uint4 t = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(row) + j);
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(w[e] << 16); v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
    } else if (j >= n_bf16) {
        const float4 a = *reinterpret_cast<const float4*>(row + j), b = *reinterpret_cast<const float4*>(row + j + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {        // the boundary group: 4 bf16 columns, then 4 fp32 columns
        const uint2 t = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(row) + j);
        const float4 b = *reinterpret_cast<const float4*>(row + j + 4);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
}

__global__ __launch_bounds__(256) void reduce_deferred_kernel(const DeferTable t) {
    __shared__ float4 red[DF_TY][DF_TX];
    int s = 0;
    while (s + 1 < t.n_seg && (int)blockIdx.x >= t.s[s + 1].first_block) ++s;
    const float* __restrict__ part = t.s[s].part;
    const long long stride = t.s[s].stride;
    const int n = t.s[s].n, n_bf16 = t.s[s].n_bf16, P = t.s[s].P;
    if (t.s[s].wide == 1) {          // (uniform over the workgroup)
        const int j = (((int)blockIdx.x - t.s[s].first_block) * 256 + (int)threadIdx.x) * 8;
        if (j >= n) return;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        if (j + 8 <= n) {
            int q = 0;
            for (; q + 8 <= P; q += 8) {
                float v[8][8];
#pragma unroll
                for (int u = 0; u < 8; ++u) defer_ld8(part, stride, q + u, j, n_bf16, v[u]);
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += v[u][e];
            }
            for (; q < P; ++q) {
                float v[8];
                defer_ld8(part, stride, q, j, n_bf16, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v[e];
            }
            float4* o = reinterpret_cast<float4*>(t.s[s].out + j);
            float4 r0 = make_float4(acc[0], acc[1], acc[2], acc[3]), r1 = make_float4(acc[4], acc[5], acc[6], acc[7]);
            if (t.s[s].accumulate) {
                const float4 a = o[0], b = o[1];
                r0.x += a.x; r0.y += a.y; r0.z += a.z; r0.w += a.w; r1.x += b.x; r1.y += b.y; r1.z += b.z; r1.w += b.w;
            }
            o[0] = r0;
            o[1] = r1;
        } else {                // the last 4 columns of a segment whose width is 4 mod 8 (fp32 or bf16 alike)
            const bool bf = j < n_bf16;
            for (int q = 0; q < P; ++q) {
                if (bf) {
                    const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(part + q * stride) + j);
                    acc[0] += __uint_as_float(v.x << 16); acc[1] += __uint_as_float(v.x & 0xffff0000u);
                    acc[2] += __uint_as_float(v.y << 16); acc[3] += __uint_as_float(v.y & 0xffff0000u);
                } else {
                    const float4 v = *reinterpret_cast<const float4*>(part + q * stride + j);
                    acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
                }
            }
            float4* o = reinterpret_cast<float4*>(t.s[s].out + j);
            float4 r0 = make_float4(acc[0], acc[1], acc[2], acc[3]);
            if (t.s[s].accumulate) { const float4 a = o[0]; r0.x += a.x; r0.y += a.y; r0.z += a.z; r0.w += a.w; }
            o[0] = r0;
        }
        return;
    }
    if (t.s[s].wide == 2) {     // unaligned fp32 segments (a width that is no multiple of 4: the heads' 7 / 2827 output rows): a column per thread
        const int j = ((int)blockIdx.x - t.s[s].first_block) * 256 + (int)threadIdx.x;
        if (j >= n) return;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int q = 0;
        for (; q + 4 <= P; q += 4) {
            const float v0 = part[(q + 0) * stride + j], v1 = part[(q + 1) * stride + j];
            const float v2 = part[(q + 2) * stride + j], v3 = part[(q + 3) * stride + j];
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        for (; q < P; ++q) a0 += part[q * stride + j];
        float r = (a0 + a1) + (a2 + a3);
        if (t.s[s].accumulate) r += t.s[s].out[j];
        t.s[s].out[j] = r;
        return;
    }
    const int tx = threadIdx.x % DF_TX, ty = threadIdx.x / DF_TX;
    const int j = (((int)blockIdx.x - t.s[s].first_block) * DF_TX + tx) * 4;
    float4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < n) {
        const bool bf = j < n_bf16;
        auto ld4 = [&](int q) -> float4 {
            if (bf) {
                const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(part + q * stride) + j);
                return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                                   __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
            }
            return *reinterpret_cast<const float4*>(part + q * stride + j);
        };
        int q = ty;
        for (; q + 3 * DF_TY < P; q += 4 * DF_TY) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ld4(q + DF_TY * u);
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc[u].x += v[u].x; acc[u].y += v[u].y; acc[u].z += v[u].z; acc[u].w += v[u].w; }
        }
        for (; q < P; q += DF_TY) {
            const float4 v = ld4(q);
            acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
        }
    }
    red[ty][tx] = make_float4((acc[0].x + acc[1].x) + (acc[2].x + acc[3].x), (acc[0].y + acc[1].y) + (acc[2].y + acc[3].y),
                              (acc[0].z + acc[1].z) + (acc[2].z + acc[3].z), (acc[0].w + acc[1].w) + (acc[2].w + acc[3].w));
    __syncthreads();
    if (ty == 0 && j < n) {
        float4 r = red[0][tx];
#pragma unroll
        for (int u = 1; u < DF_TY; ++u) { const float4 v = red[u][tx]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
        float4* o = reinterpret_cast<float4*>(t.s[s].out + j);
        if (t.s[s].accumulate) { const float4 v = *o; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
        *o = r;
    }
}

int defer_flush_locked(DeferQueue& d, hipStream_t st) {
    // weight-gradient GEMMs still waiting in a group on this stream produce partials queued here: they go first
    if (int rc = dsvg_gemm_group_flush(st)) return rc;
    size_t at = 0;
    while (at < d.q.size()) {
        DeferTable t;
        int blocks = 0, k = 0;
        for (; k < DEFER_MAX && at < d.q.size(); ++k, ++at) {
            t.s[k] = d.q[at];
            t.s[k].first_block = blocks;
            // wide layout: 16-byte accesses of 8 columns (slices and destination 16-byte aligned, the bf16 / fp32 boundary on a
            // group boundary); few slices, many columns
            const bool vec = !(t.s[k].n & 3) && !(t.s[k].stride & 3) && !((uintptr_t)t.s[k].part & 15) &&
                             !((uintptr_t)t.s[k].out & 15) && !(t.s[k].n_bf16 & 3);
            t.s[k].wide = !vec ? 2 : (t.s[k].n >= DF_WIDE_COLS && t.s[k].P <= 64 && !(t.s[k].n_bf16 & 7)) ? 1 : 0;
            blocks += t.s[k].wide == 2 ? dsvg_cdiv(t.s[k].n, 256) : t.s[k].wide ? dsvg_cdiv(t.s[k].n, DF_WIDE_COLS) : dsvg_cdiv(t.s[k].n, DF_TX * 4);
        }
        t.n_seg = k;
        hipLaunchKernelGGL(reduce_deferred_kernel, dim3(blocks), dim3(256), 0, st, t);
    }
    d.q.clear();
    DSVG_LAUNCH_CHECK("reduce_deferred");
    return 0;
}

// true when [out, out + n) overlaps the destination of a queued segment (the queue must then be flushed first)
bool defer_overlaps(const DeferQueue& d, const float* out, int64_t n) {
    for (const DeferSeg& g : d.q)
        if (out < g.out + g.n && g.out < out + n) return true;
    return false;
}
}  // namespace

extern "C" int dsvg_defer_scope(int32_t on, void* stream) {
    DeferTableOfQueues& t = defer_queues();
    std::lock_guard<std::mutex> lk(t.mu);
    auto it = t.by_stream.find((hipStream_t)stream);
    if (it == t.by_stream.end()) {
        if (!on) return 0;
        it = t.by_stream.emplace((hipStream_t)stream, DeferQueue{}).first;
    }
    it->second.scope = on ? 1 : 0;
    const int queued = (int)it->second.q.size();
    if (!on && queued == 0) t.by_stream.erase(it);      // closed and drained: forget the stream
    return queued;
}

extern "C" int dsvg_flush_deferred(void* stream) {
    DeferTableOfQueues& t = defer_queues();
    std::lock_guard<std::mutex> lk(t.mu);
    auto it = t.by_stream.find((hipStream_t)stream);
    if (it == t.by_stream.end() || it->second.q.empty()) return 0;
    // (the queued partials were produced on this very stream: the launch below is ordered behind all of them)
    const int rc = defer_flush_locked(it->second, (hipStream_t)stream);
    if (!it->second.scope) t.by_stream.erase(it);
    return rc;
}

extern "C" int dsvg_defer_zero(float* out, int64_t n, void* stream) {
    if (n <= 0) return 0;
    DSVG_CHECK_ARG(out && n < (1ll << 30), "defer_zero: bad args");
    hipStream_t st = (hipStream_t)stream;
    {
        DeferTableOfQueues& t = defer_queues();
        std::lock_guard<std::mutex> lk(t.mu);
        auto it = t.by_stream.find(st);
        if (it != t.by_stream.end()) {
            DeferQueue& d = it->second;
            if (!d.q.empty() && defer_overlaps(d, out, n)) {
                int rc = defer_flush_locked(d, st);
                if (rc) return rc;
            }
            if (d.scope) {      // a reduction over ZERO partial rows writes zeros: the fill rides on the flush's launches
                d.q.push_back(DeferSeg{out, out, 0ll, (int)n, 0, 0, 0, 0, 0});
                return 0;
            }
        }
    }
    if (int rc = dsvg_gemm_group_flush(st)) return rc;
    if (hipMemsetAsync(out, 0, (size_t)n * sizeof(float), st) != hipSuccess) { dsvg_set_error("defer_zero: hipMemsetAsync failed"); return -1; }
    return 0;
}

int dsvg_reduce_partials_mixed(const float* part, int64_t P, int64_t stride, int64_t n, int64_t n_bf16, float* out,
                               int32_t accumulate, hipStream_t st) {
    if (n <= 0) return 0;
    const bool vec = !(n & 3) && !(stride & 3) && !((uintptr_t)part & 15) && !((uintptr_t)out & 15) && !(n_bf16 & 3);
    {
        DeferTableOfQueues& t = defer_queues();
        std::lock_guard<std::mutex> lk(t.mu);
        auto it = t.by_stream.find(st);
        if (it != t.by_stream.end()) {
            DeferQueue& d = it->second;
            if (!d.q.empty() && defer_overlaps(d, out, n)) {
                // a second write to a queued destination: order matters, run the queue now (same stream: ordered)
                int rc = defer_flush_locked(d, st);
                if (rc) return rc;
            }
            // (segments that are no 16-byte multiples - the heads' 7 / 2827 output rows - take the kernel's column-per-thread path;
            // DSVG_DEFER_MORE=0: reduced on the spot as before round 5, an A/B knob)
            static const bool unaligned_ok = !(getenv("DSVG_DEFER_MORE") && atoi(getenv("DSVG_DEFER_MORE")) == 0);
            if (d.scope && (vec || (unaligned_ok && n_bf16 == 0)) && n < (1ll << 30) && P < (1ll << 30)) {
                d.q.push_back(DeferSeg{part, out, (long long)stride, (int)n, (int)n_bf16, (int)P, 0, accumulate, 0});
                return 0;
            }
        }
    }
    if (int rc = dsvg_gemm_group_flush(st)) return rc;      // (an immediate reduction must not overtake a still-queued producer)
    if (n_bf16 > 0 && !vec) { dsvg_set_error("reduce_partials: bf16 partials need 16-byte aligned, 4-column-multiple data"); return -1; }
    const long long cols = vec ? n / 4 : n;
    // few columns + many partial rows (bias / LayerNorm gradients): 16 slices over the partial index per block;
    // many columns (weight gradients): 64 column lanes x 4 slices
    const bool wide = cols >= 64 * 128 || P <= 16;
#define DSVG_RP(V, X, B) hipLaunchKernelGGL((reduce_partials_strided_kernel<V, X, B>), dim3(dsvg_cdiv(cols, X)), dim3(256), 0, \
                                            st, part, (long long)P, (long long)stride, (long long)n, out, accumulate,      \
                                            (long long)n_bf16)
    // a handful of columns (the loss sums: n = 2 from up to 2048 partial rows): 64 slices over the partial index - with 16
    // of them two threads per slice summed 128 rows each, a 10 us chain of dependent loads
    const bool narrow = !vec && n_bf16 == 0 && cols <= 4 && P > 64;
    if (n_bf16 > 0) { if (wide) DSVG_RP(4, 64, true); else DSVG_RP(4, 16, true); }
    else if (vec) { if (wide) DSVG_RP(4, 64, false); else DSVG_RP(4, 16, false); }
    else if (narrow) DSVG_RP(1, 4, false);
    else     { if (wide) DSVG_RP(1, 64, false); else DSVG_RP(1, 16, false); }
#undef DSVG_RP
    DSVG_LAUNCH_CHECK("reduce_partials");
    return 0;
}

// column sums of a [M, N] matrix (bias gradients).  A block owns CS_ROWS rows and a panel of up to 64 column
// chunks (16 bytes each); its 256 threads are (256/cpb) row groups x cpb column chunks, every thread streams
// 16-byte row segments with independent accumulators; row groups are combined through LDS; one partial row
// per block goes to the workspace and is reduced in a fixed order.
constexpr int CS_ROWS_MIN = 16, CS_ROWS_MAX = 1024;      // (round 6: 16 instead of 64 - a 4096-row column sum was 64 workgroups walking 8 dependent row loads each: 12-16 us)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ A, long long lda, long long M, int N,
                                                     float* __restrict__ part, float drop_p, uint32_t site,
                                                     const uint64_t* seed, int cpb, int vec_ok, int rows_per_block) {
    constexpr int VEC = 16 / (int)sizeof(T);
    __shared__ float red[256][VEC + 1];
    const DropCtx dc = drop_make(drop_p, seed, site);
    const int cl = threadIdx.x % cpb, rg = threadIdx.x / cpb, nrg = 256 / cpb;
    const int c0 = (blockIdx.y * cpb + cl) * VEC;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(M, r0 + rows_per_block);
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    if (c0 < N) {
        if (vec_ok && c0 + VEC <= N) {
            for (long long r = r0 + rg; r < r1; r += nrg) {
                float v[VEC];
                if (VEC == 4) {
                    float t[4];
                    Elem<T>::ld4(A + r * lda + c0, t);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e % VEC] = t[e];
                } else {
                    float t0[4], t1[4];
                    Elem<T>::ld4(A + r * lda + c0, t0);
                    Elem<T>::ld4(A + r * lda + c0 + 4, t1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = t0[e]; v[(e + 4) % VEC] = t1[e]; }
                }
                if (VEC == 8 && dc.on && !(N & 7)) {
                    float m8[8];
                    drop_mult8(dc, (uint64_t)r * N + c0, m8);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] += v[e] * m8[e % 8];
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] += v[e] * drop_mult(dc, (uint64_t)r * N + c0 + e);
                }
            }
        } else {
            for (long long r = r0 + rg; r < r1; r += nrg)
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (c0 + e < N) acc[e] += Elem<T>::ld(A + r * lda + c0 + e) * drop_mult(dc, (uint64_t)r * N + c0 + e);
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[threadIdx.x][e] = acc[e];
    __syncthreads();
    if (rg == 0 && c0 < N) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            if (c0 + e < N) {
                float s = 0.f;
                for (int g = 0; g < nrg; ++g) s += red[g * cpb + cl][e];
                part[(long long)blockIdx.x * N + c0 + e] = s;
            }
        }
    }
}

extern "C" int64_t dsvg_gemm_workspace_bytes(int32_t M, int32_t N, int32_t split_k) {
    if (split_k <= 1) return 0;
    return (int64_t)split_k * (int64_t)dsvg_splitk_slice(M, N, true) * (int64_t)sizeof(float);   // [split][M*N | M row sums]
}

extern "C" int dsvg_reduce_partials(const float* partial, int64_t P, int64_t n, float* out,
                                    int32_t accumulate, void* stream) {
    DSVG_CHECK_ARG(partial && out && P >= 0 && n >= 0, "reduce_partials: bad args");
    return dsvg_reduce_partials_strided(partial, P, n, n, out, accumulate, (hipStream_t)stream);
}

extern "C" int64_t dsvg_colsum_workspace_bytes(int64_t M, int32_t N) {
    return (int64_t)dsvg_cdiv(M, CS_ROWS_MIN) * N * (int64_t)sizeof(float);
}

extern "C" int dsvg_colsum(int32_t dtype, const void* A, int64_t lda, int64_t M, int32_t N, float* out,
                           int32_t accumulate, float drop_p, uint32_t drop_site, const uint64_t* seed,
                           float* workspace, int64_t workspace_bytes, void* stream) {
    DSVG_CHECK_ARG(A && out && M > 0 && N > 0, "colsum: bad args");
    DSVG_CHECK_ARG(workspace && workspace_bytes >= dsvg_colsum_workspace_bytes(M, N),
                   "colsum: workspace too small (%lld < %lld)", (long long)workspace_bytes,
                   (long long)dsvg_colsum_workspace_bytes(M, N));
    hipStream_t st = (hipStream_t)stream;
    DSVG_CHECK_ARG(dtype == DSVG_F32 || dtype == DSVG_BF16, "colsum: bad dtype %d", dtype);
    const int vec = dtype == DSVG_F32 ? 4 : 8;
    const int chunks = dsvg_cdiv(N, vec);
    int cpb = 8;
    while (cpb < 64 && cpb < chunks) cpb *= 2;
    const int vec_ok = !(lda % vec) && !((uintptr_t)A & 15);
    const int col_blocks = dsvg_cdiv(chunks, cpb);
    // enough row blocks to fill the chip (~1024 workgroups), between 16 and 1024 rows each
    long long rpb = M / max(1, 1024 / col_blocks);
    rpb = rpb < CS_ROWS_MIN ? CS_ROWS_MIN : (rpb > CS_ROWS_MAX ? CS_ROWS_MAX : rpb);
    const int nb = dsvg_cdiv(M, rpb);
    dim3 grid(nb, col_blocks);
    if (dtype == DSVG_F32)
        hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, st, (const float*)A, (long long)lda,
                           (long long)M, N, workspace, drop_p, drop_site, seed, cpb, vec_ok, (int)rpb);
    else
        hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)A, (long long)lda,
                           (long long)M, N, workspace, drop_p, drop_site, seed, cpb, vec_ok, (int)rpb);
    DSVG_LAUNCH_CHECK("colsum");
    return dsvg_reduce_partials(workspace, nb, N, out, accumulate, stream);
}

int dsvg_gemm_bf16_launch(const dsvg_gemm_desc& d, int k_chunk, int nsplit, float* part, float* rs_part,
                          hipStream_t st, int* part_is_bf16);  // gemm_bf16.hip

extern "C" int dsvg_gemm(const dsvg_gemm_desc* dp, void* stream) {
    DSVG_CHECK_ARG(dp, "gemm: null desc");
    dsvg_gemm_desc d = *dp;
    hipStream_t st = (hipStream_t)stream;
    DSVG_CHECK_ARG(d.dtype == DSVG_F32 || d.dtype == DSVG_BF16, "gemm: bad dtype %d", d.dtype);
    DSVG_CHECK_ARG(d.M > 0 && d.N > 0 && d.K > 0, "gemm: bad shape %d %d %d", d.M, d.N, d.K);
    DSVG_CHECK_ARG(d.A && d.B && d.C, "gemm: null operand");
    DSVG_CHECK_ARG((d.drop_p <= 0.f && d.a_drop_p <= 0.f) || d.seed, "gemm: dropout needs a seed pointer");
    if (d.dtype == DSVG_F32) d.c_f32 = 1;
    const int split = d.split_k > 1 ? d.split_k : 1;
    float* part = nullptr;
    int k_chunk = d.K;
    if (split > 1) {
        DSVG_CHECK_ARG(d.c_f32, "gemm: split_k needs fp32 output");
        DSVG_CHECK_ARG(!d.bias && !d.res && !d.gate && d.act == 0 && d.drop_p <= 0.f,
                       "gemm: split_k does not support an epilogue");
        DSVG_CHECK_ARG(d.workspace && d.workspace_bytes >= dsvg_gemm_workspace_bytes(d.M, d.N, split),
                       "gemm: split_k workspace too small");
        DSVG_CHECK_ARG(d.ldc == d.N, "gemm: split_k needs a dense C (ldc == N)");
        part = d.workspace;
        k_chunk = ((d.K + split - 1) / split + 63) / 64 * 64;
    }
    int nsplit = (d.K + k_chunk - 1) / k_chunk;
    if (split > 1 && (split % 8) == 0) nsplit = split;   // trailing slices may be empty (they write zero partials)
    DSVG_CHECK_ARG(!d.rowsum || part, "gemm: rowsum needs split_k > 1");
    float* rs_part = d.rowsum ? part + (size_t)d.M * d.N : nullptr;   // row sums of slice 0 (slices are interleaved)
    const size_t slice = dsvg_splitk_slice(d.M, d.N, d.rowsum != nullptr);

    int part_bf16 = 0;
    bool use_naive = d.impl == 1;
    if (d.dtype == DSVG_F32) {
        // the MFMA kernel needs 16-byte aligned rows
        if ((d.lda & 3) || (d.ldb & 3) || ((uintptr_t)d.A & 15) || ((uintptr_t)d.B & 15)) use_naive = true;
    } else {
        const int Kp = (d.K + 7) / 8 * 8;
        if ((d.lda & 7) || (d.ldb & 7) || ((uintptr_t)d.A & 15) || ((uintptr_t)d.B & 15)) use_naive = true;
        if ((d.a_kc && d.lda < Kp) || (d.b_kc && d.ldb < Kp)) use_naive = true;
        if (!d.a_kc && d.lda < ((d.M + 7) / 8) * 8) use_naive = true;
        if (!d.b_kc && d.ldb < ((d.N + 7) / 8) * 8) use_naive = true;
    }
    if (use_naive) {
        const long long total = (long long)d.M * d.N;
        for (int z = 0; z < nsplit; ++z) {
            const int kb = z * k_chunk, ke = min(d.K, kb + k_chunk);
            float* pz = part ? part + (size_t)z * slice : nullptr;
            float* rz = rs_part ? rs_part + (size_t)z * slice : nullptr;
            if (d.dtype == DSVG_F32)
                hipLaunchKernelGGL(gemm_naive_kernel<float>, dim3(dsvg_cdiv(total, 256)), dim3(256), 0, st, d, kb, ke, pz, rz);
            else
                hipLaunchKernelGGL(gemm_naive_kernel<bf16_t>, dim3(dsvg_cdiv(total, 256)), dim3(256), 0, st, d, kb, ke, pz, rz);
        }
        DSVG_LAUNCH_CHECK("gemm_naive");
    } else if (d.dtype == DSVG_F32) {
        const int tiles_m = dsvg_cdiv(d.M, BM), tiles_n = dsvg_cdiv(d.N, BN);
        const int nwg = tiles_m * tiles_n;
        dim3 grid(nwg, nsplit);
        if (nsplit > 1 && (nsplit % 8) == 0) grid = dim3(nwg * nsplit, 1);
#define DSVG_F32V(AK, BK)                                                                                              \
    do {                                                                                                               \
        if (rs_part) hipLaunchKernelGGL((gemm_f32_mfma_kernel<AK, BK, true>), grid, dim3(256), 0, st, d, tiles_n, nwg, \
                                        k_chunk, part, rs_part);                                                       \
        else hipLaunchKernelGGL((gemm_f32_mfma_kernel<AK, BK, false>), grid, dim3(256), 0, st, d, tiles_n, nwg,        \
                                k_chunk, part, rs_part);                                                               \
    } while (0)
        if (d.a_kc && d.b_kc) DSVG_F32V(true, true);
        else if (d.a_kc && !d.b_kc) DSVG_F32V(true, false);
        else if (!d.a_kc && d.b_kc) DSVG_F32V(false, true);
        else DSVG_F32V(false, false);
#undef DSVG_F32V
        DSVG_LAUNCH_CHECK("gemm_f32_mfma");
    } else {
        int rc = dsvg_gemm_bf16_launch(d, k_chunk, nsplit, part, rs_part, st, &part_bf16);
        if (rc) return rc;
    }
    if (part && part_bf16) {    // bf16 partials (fp32 row sums behind them at their usual float index): one mixed reduction
        const int64_t mn = (int64_t)d.M * d.N;
        if (rs_part && d.rowsum == (float*)d.C + mn && !(d.M & 3))
            return dsvg_reduce_partials_mixed(part, nsplit, (int64_t)slice, mn + d.M, mn, (float*)d.C, d.accumulate, st);
        int rc = dsvg_reduce_partials_mixed(part, nsplit, (int64_t)slice, mn, mn, (float*)d.C, d.accumulate, st);
        if (rc) return rc;
        if (rs_part) return dsvg_reduce_partials_strided(rs_part, nsplit, (int64_t)slice, d.M, d.rowsum, d.accumulate, st);
        return 0;
    }
    if (part) {
        const int64_t mn = (int64_t)d.M * d.N;
        // dW and db are adjacent in the flat gradient buffer (bias follows weight): one reduction covers both
        if (rs_part && d.rowsum == (float*)d.C + mn)
            return dsvg_reduce_partials_strided(part, nsplit, (int64_t)slice, mn + d.M, (float*)d.C, d.accumulate, st);
        int rc = dsvg_reduce_partials_strided(part, nsplit, (int64_t)slice, mn, (float*)d.C, d.accumulate, st);
        if (rc) return rc;
        if (rs_part) return dsvg_reduce_partials_strided(rs_part, nsplit, (int64_t)slice, d.M, d.rowsum, d.accumulate, st);
    }
    return 0;
}
