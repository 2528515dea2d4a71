// LayerNorm forward / backward over rows of d features (d % 4 == 0, d <= 1024), one 64-lane wave per
// row, 4 rows per 256-thread block, grid-stride over rows.  HBM-bound: every element is read once with
// 16-byte (fp32) / 8-byte (bf16) lane accesses and kept in registers between the two passes.
// Replaces torch.nn.LayerNorm at deepsvg/model/layers/improved_transformer.py:43,51,127,138 and
// deepsvg/model/layers/transformer.py:185-186,239-240, plus native_layer_norm_backward.
#include "dsvg_common.h"
#include "../../include/dsvg.h"

constexpr int LN_MAXV_MAX = 4;        // 4 x (64 lanes x 4 elements) = 1024 features max; kernels are built for 1, 2, 4
constexpr int LN_MAX_BLOCKS = 1024;   // backward: one partial row of the parameter gradients per workgroup (512..4096 measured equal)

template <typename T, int LN_MAXV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     long long rows, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nv = d / 256 + ((d % 256) ? 1 : 0);
    // row-invariant scale / shift: loaded once, up front and unconditionally (clamped column) so that they are in
    // flight together with the row itself instead of costing a dependent round trip after the reductions
    float4 g4[LN_MAXV], b4[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = min(i * 256 + lane * 4, d - 4);
        g4[i] = *reinterpret_cast<const float4*>(gamma + c);
        b4[i] = *reinterpret_cast<const float4*>(beta + c);
    }
    for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
        const T* xr = x + row * d;
        float v[LN_MAXV][4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = i * 256 + lane * 4;
            if (i < nv && c < d) {
                Elem<T>::ld4(xr + c, v[i]);
                s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
            } else {
                v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
            }
        }
        const float mu = wave_sum(s) / (float)d;
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = i * 256 + lane * 4;
            if (i < nv && c < d) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float t = v[i][e] - mu; s2 += t * t; }
            }
        }
        const float var = wave_sum(s2) / (float)d;
        const float rs = rsqrtf(var + eps);
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
        T* yr = y + row * d;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = i * 256 + lane * 4;
            if (i < nv && c < d) {
                const float4 g = g4[i];
                const float4 b = b4[i];
                float o[4];
                o[0] = (v[i][0] - mu) * rs * g.x + b.x;
                o[1] = (v[i][1] - mu) * rs * g.y + b.y;
                o[2] = (v[i][2] - mu) * rs * g.z + b.z;
                o[3] = (v[i][3] - mu) * rs * g.w + b.w;
                Elem<T>::st4(yr + c, o);
            }
        }
    }
}

// dx = [res +] rstd * (g*dy - mean_d(g*dy) - xhat * mean_d(g*dy*xhat))
// dgamma/dbeta: per-wave register accumulation over the wave's rows, combined per block in LDS, one
// partial row per block -> workspace [gridDim.x][2][d]; reduced by dsvg_reduce_partials.
template <typename T, int LN_MAXV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* dy, const T* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const T* res,
                                                     T* dx, float* __restrict__ part,
                                                     long long rows, int d, T* dxm, float drop_p,
                                                     const uint64_t* __restrict__ seed, uint32_t site) {
    // dxm (optional): a second output, dx as stored (rounded to T) with the dropout mask of `site` replayed on it - what
    // dsvg_drop_apply would make of dx: the consumer of this gradient (the FFN half of the layer below multiplies it with the
    // mask of its residual dropout) then needs no launch of its own for it (round 5)
    const DropCtx dmc = drop_make(dxm ? drop_p : 0.f, seed, site);
    // [wave][dg/db][d/4 + 1 vectors][4]: sized by d at launch (8 KiB at d = 256) so that LDS does not cap the number of
    // resident workgroups - the kernel hides HBM latency with waves, each wave walks its rows one after the other
    extern __shared__ float red_raw[];
    const int nvec = d / 4 + 1;
    auto red = [&](int w, int k, int vec, int e) -> float& { return red_raw[((w * 2 + k) * nvec + vec) * 4 + e]; };
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nv = d / 256 + ((d % 256) ? 1 : 0);
    float ag[LN_MAXV][4], ab[LN_MAXV][4];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { ag[i][e] = 0.f; ab[i][e] = 0.f; }

    float4 g4[LN_MAXV];             // row-invariant, loaded once (clamped column)
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) g4[i] = *reinterpret_cast<const float4*>(gamma + min(i * 256 + lane * 4, d - 4));
    for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        float xh[LN_MAXV][4], gd[LN_MAXV][4], rv[LN_MAXV][4];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = i * 256 + lane * 4;
            if (i < nv && c < d) {
                float xv[4], dv[4];
                Elem<T>::ld4(x + row * d + c, xv);
                Elem<T>::ld4(dy + row * d + c, dv);
                // the residual gradient is fetched with x and dy (one HBM round trip per row instead of two)
                if (res) Elem<T>::ld4(res + row * d + c, rv[i]);
                const float4 g = g4[i];
                const float gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[i][e] = (xv[e] - mu) * rs;
                    gd[i][e] = gg[e] * dv[e];
                    c1 += gd[i][e];
                    c2 += gd[i][e] * xh[i][e];
                    ag[i][e] += dv[e] * xh[i][e];
                    ab[i][e] += dv[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { xh[i][e] = 0.f; gd[i][e] = 0.f; }
            }
        }
        c1 = wave_sum(c1) / (float)d;
        c2 = wave_sum(c2) / (float)d;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = i * 256 + lane * 4;
            if (i < nv && c < d) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (gd[i][e] - c1 - xh[i][e] * c2);
                if (res) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += rv[i][e];
                }
                Elem<T>::st4(dx + row * d + c, o);
                if (dxm) {
                    // element ids row * d + c + e (groups of 8: this lane's 4 are one half of a group)
                    const uint64_t id4 = (uint64_t)row * (uint64_t)d + (uint64_t)c;
                    float mm[4] = {1.f, 1.f, 1.f, 1.f};
                    if (dmc.on) {
                        const uint32_t hg = drop_group(dmc, id4 >> 3);
                        const uint32_t wb = ((uint32_t)id4 & 4u) ? 2u : 0u;
                        const uint32_t w0 = drop_word(hg, wb), w1 = drop_word(hg, wb + 1u);
                        mm[0] = (w0 & 0xffffu) < dmc.thresh ? 0.f : dmc.scale;
                        mm[1] = (w0 >> 16) < dmc.thresh ? 0.f : dmc.scale;
                        mm[2] = (w1 & 0xffffu) < dmc.thresh ? 0.f : dmc.scale;
                        mm[3] = (w1 >> 16) < dmc.thresh ? 0.f : dmc.scale;
                    }
                    float om[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        T t;
                        Elem<T>::st(&t, o[e]);                      // the value as dx holds it
                        om[e] = Elem<T>::ld(&t) * mm[e];
                    }
                    Elem<T>::st4(dxm + row * d + c, om);
                }
            }
        }
    }
    // block combine of the parameter-gradient partials
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        if (i < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (i * 256 + lane * 4 < d) {
                    red(wave, 0, i * 64 + lane, e) = ag[i][e];
                    red(wave, 1, i * 64 + lane, e) = ab[i][e];
                }
            }
        }
    }
    __syncthreads();
    float* pg = part + (size_t)blockIdx.x * 2 * d;
    for (int c = threadIdx.x; c < d; c += 256) {
        const int vec = c >> 2, e = c & 3;
        float sg = 0.f, sb = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { sg += red(w, 0, vec, e); sb += red(w, 1, vec, e); }
        pg[c] = sg;
        pg[d + c] = sb;
    }
}

static int ln_grid(long long rows) {
    static const int cap = getenv("DSVG_LN_BLOCKS") ? atoi(getenv("DSVG_LN_BLOCKS")) : LN_MAX_BLOCKS;   // tuning knob
    long long nb = (rows + 3) / 4;
    return (int)(nb < cap ? nb : cap);
}
// forward: no per-block partials, so one wave per row with every row's load in flight at once (a wave that walks 30
// rows one after the other is a chain of dependent ~2 us loads: measured 58 us for 134 MB of traffic)
static int ln_grid_fwd(long long rows) {
    long long nb = (rows + 3) / 4;
    return (int)(nb < 65536 ? nb : 65536);
}

extern "C" int dsvg_layernorm_fwd(int32_t dtype, const void* x, const float* gamma, const float* beta, void* y,
                                  float* mean, float* rstd, int64_t rows, int32_t d, float eps, void* stream) {
    DSVG_CHECK_ARG(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null pointer");
    DSVG_CHECK_ARG(rows > 0 && d > 0 && (d % 4) == 0 && d <= 1024, "layernorm_fwd: bad shape rows=%lld d=%d",
                   (long long)rows, d);
    hipStream_t st = (hipStream_t)stream;
#define DSVG_LNF(TT, NV) hipLaunchKernelGGL((ln_fwd_kernel<TT, NV>), dim3(ln_grid_fwd(rows)), dim3(256), 0, st, \
                                            (const TT*)x, gamma, beta, (TT*)y, mean, rstd, (long long)rows, d, eps)
    if (dtype == DSVG_F32) { if (d <= 256) DSVG_LNF(float, 1); else if (d <= 512) DSVG_LNF(float, 2); else DSVG_LNF(float, 4); }
    else if (dtype == DSVG_BF16) { if (d <= 256) DSVG_LNF(bf16_t, 1); else if (d <= 512) DSVG_LNF(bf16_t, 2); else DSVG_LNF(bf16_t, 4); }
    else { dsvg_set_error("layernorm_fwd: bad dtype %d", dtype); return -1; }
#undef DSVG_LNF
    DSVG_LAUNCH_CHECK("layernorm_fwd");
    return 0;
}

extern "C" int64_t dsvg_layernorm_bwd_workspace_bytes(int64_t rows, int32_t d) {
    return (int64_t)ln_grid(rows) * 2 * d * (int64_t)sizeof(float);
}

static int ln_bwd_launch(int32_t dtype, const void* dy, const void* x, const float* mean, const float* rstd,
                         const float* gamma, const void* res, void* dx, float* dgamma, float* dbeta,
                         int32_t accumulate, int64_t rows, int32_t d, float* workspace,
                         int64_t workspace_bytes, void* dx_masked, float drop_p, uint32_t drop_site, const void* seed,
                         void* stream);

extern "C" int dsvg_layernorm_bwd(int32_t dtype, const void* dy, const void* x, const float* mean, const float* rstd,
                                  const float* gamma, const void* res, void* dx, float* dgamma, float* dbeta,
                                  int32_t accumulate, int64_t rows, int32_t d, float* workspace,
                                  int64_t workspace_bytes, void* stream) {
    return ln_bwd_launch(dtype, dy, x, mean, rstd, gamma, res, dx, dgamma, dbeta, accumulate, rows, d, workspace,
                         workspace_bytes, nullptr, 0.f, 0u, nullptr, stream);
}

extern "C" int dsvg_layernorm_bwd_masked(int32_t dtype, const void* dy, const void* x, const float* mean, const float* rstd,
                                         const float* gamma, const void* res, void* dx, float* dgamma, float* dbeta,
                                         int32_t accumulate, int64_t rows, int32_t d, float* workspace,
                                         int64_t workspace_bytes, void* dx_masked, float drop_p, uint32_t drop_site,
                                         const void* seed, void* stream) {
    DSVG_CHECK_ARG(dx_masked && dx_masked != dx, "layernorm_bwd_masked: needs a second output buffer");
    DSVG_CHECK_ARG(!(drop_p > 0.f) || seed, "layernorm_bwd_masked: dropout needs a seed");
    return ln_bwd_launch(dtype, dy, x, mean, rstd, gamma, res, dx, dgamma, dbeta, accumulate, rows, d, workspace,
                         workspace_bytes, dx_masked, drop_p, drop_site, seed, stream);
}

static int ln_bwd_launch(int32_t dtype, const void* dy, const void* x, const float* mean, const float* rstd,
                         const float* gamma, const void* res, void* dx, float* dgamma, float* dbeta,
                         int32_t accumulate, int64_t rows, int32_t d, float* workspace,
                         int64_t workspace_bytes, void* dx_masked, float drop_p, uint32_t drop_site, const void* seed,
                         void* stream) {
    DSVG_CHECK_ARG(dy && x && mean && rstd && gamma && dx && dgamma && dbeta, "layernorm_bwd: null pointer");
    DSVG_CHECK_ARG(rows > 0 && d > 0 && (d % 4) == 0 && d <= 1024, "layernorm_bwd: bad shape");
    DSVG_CHECK_ARG(workspace && workspace_bytes >= dsvg_layernorm_bwd_workspace_bytes(rows, d),
                   "layernorm_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int nb = ln_grid(rows);
    const size_t lds = (size_t)4 * 2 * (d / 4 + 1) * 4 * sizeof(float);
#define DSVG_LNB(TT, NV) hipLaunchKernelGGL((ln_bwd_kernel<TT, NV>), dim3(nb), dim3(256), lds, st, (const TT*)dy, \
                                            (const TT*)x, mean, rstd, gamma, (const TT*)res, (TT*)dx, workspace,  \
                                            (long long)rows, d, (TT*)dx_masked, drop_p, (const uint64_t*)seed, drop_site)
    if (dtype == DSVG_F32) { if (d <= 256) DSVG_LNB(float, 1); else if (d <= 512) DSVG_LNB(float, 2); else DSVG_LNB(float, 4); }
    else if (dtype == DSVG_BF16) { if (d <= 256) DSVG_LNB(bf16_t, 1); else if (d <= 512) DSVG_LNB(bf16_t, 2); else DSVG_LNB(bf16_t, 4); }
    else { dsvg_set_error("layernorm_bwd: bad dtype %d", dtype); return -1; }
#undef DSVG_LNB
    DSVG_LAUNCH_CHECK("layernorm_bwd");
    // workspace rows are [dgamma(d) | dbeta(d)]; in the flat gradient buffer norm.bias follows norm.weight, so the
    // usual case is ONE deterministic reduction of 2d columns, otherwise two strided ones
    if (dbeta == dgamma + d)
        return dsvg_reduce_partials_strided(workspace, nb, 2 * (int64_t)d, 2 * (int64_t)d, dgamma, accumulate, st);
    int rc = 0;
    rc = dsvg_reduce_partials_strided(workspace, nb, 2 * (int64_t)d, d, dgamma, accumulate, st);
    if (rc) return rc;
    rc = dsvg_reduce_partials_strided(workspace + d, nb, 2 * (int64_t)d, d, dbeta, accumulate, st);
    return rc;
}
