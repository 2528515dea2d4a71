// Flat-buffer optimizer step for the data-parallel trainer: global gradient norm, clip + AdamW in one
// pass over one contiguous fp32 parameter buffer (28 B/param of HBM traffic, one launch instead of the
// 244-tensor foreach of deepsvg/train.py:100-102), plus the per-step bf16 weight image and the dropout
// seed advance.  Everything reads its scalars (lr, step, norm) from device memory so the whole step can
// be replayed from a captured hipGraph.
#include "dsvg_common.h"
#include "pack_images.h"
#include "../../include/dsvg.h"

constexpr int SQ_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long long n, float* __restrict__ part) {
    __shared__ float red[4];
    float s = 0.f;
    const long long n4 = n / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0)
        for (long long i = n4 * 4 + threadIdx.x; i < n; i += 256) s += x[i] * x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

extern "C" int64_t dsvg_sumsq_workspace_bytes(int64_t n) { (void)n; return SQ_BLOCKS * (int64_t)sizeof(float); }

extern "C" int dsvg_sumsq(const float* x, int64_t n, float* out, float* workspace, int64_t workspace_bytes, void* stream) {
    DSVG_CHECK_ARG(x && out && n > 0, "sumsq: bad args");
    DSVG_CHECK_ARG(((uintptr_t)x & 15) == 0, "sumsq: buffer must be 16-byte aligned");
    DSVG_CHECK_ARG(workspace && workspace_bytes >= dsvg_sumsq_workspace_bytes(n), "sumsq: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)min((long long)SQ_BLOCKS, (long long)dsvg_cdiv(n, 1024));
    hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, st, x, (long long)n, workspace);
    DSVG_LAUNCH_CHECK("sumsq");
    return dsvg_reduce_partials_strided(workspace, nb, 1, 1, out, 0, st);
}

// torch.nn.utils.clip_grad_norm_ : coef = min(1, max_norm / (norm + 1e-6));  g *= coef
// torch.optim.AdamW (amsgrad=False, maximize=False):
//   p *= 1 - lr*wd ; m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
//   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long long n,
                                                    const float* __restrict__ lr_p, float b1, float b2, float eps,
                                                    float wd, const long long* __restrict__ step_p,
                                                    const float* __restrict__ gnorm_sq, float max_norm,
                                                    float grad_scale) {
    const float lr = *lr_p;
    const float t = (float)(*step_p);
    float coef = grad_scale;
    if (gnorm_sq && max_norm > 0.f) {
        const float norm = sqrtf(*gnorm_sq) * grad_scale;
        coef *= fminf(1.f, max_norm / (norm + 1e-6f));
    }
    const float bc1 = 1.f - powf(b1, t);
    const float bc2s = sqrtf(1.f - powf(b2, t));
    const float step_size = lr / bc1;
    const float decay = 1.f - lr * wd;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i] * coef;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2s + eps;
        p[i] = p[i] * decay - step_size * (mi / denom);
    }
}

extern "C" int dsvg_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, const float* lr, float beta1,
                               float beta2, float eps, float weight_decay, const int64_t* step, const float* gnorm_sq,
                               float max_norm, float grad_scale, void* stream) {
    DSVG_CHECK_ARG(p && g && m && v && lr && step && n > 0, "adamw_step: bad args");
    const int nb = (int)min(4096LL, (long long)dsvg_cdiv(n, 256));
    hipLaunchKernelGGL(adamw_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, lr, beta1,
                       beta2, eps, weight_decay, (const long long*)step, gnorm_sq, max_norm, grad_scale);
    DSVG_LAUNCH_CHECK("adamw_step");
    return 0;
}

// dst = cast(src); dst_t = cast(src)^T  (bf16 weight images for the MFMA GEMMs, rebuilt every step)
template <typename T>
__global__ void cast_weights_kernel(const float* __restrict__ src, T* __restrict__ dst, T* __restrict__ dst_t,
                                    long long rows, long long cols) {
    __shared__ float tile[32][33];
    const long long c0 = (long long)blockIdx.x * 32, r0 = (long long)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const long long r = r0 + j, c = c0 + tx;
        float v = 0.f;
        if (r < rows && c < cols) {
            v = src[r * cols + c];
            if (dst) Elem<T>::st(dst + r * cols + c, v);
        }
        tile[j][tx] = v;
    }
    __syncthreads();
    if (dst_t) {
        for (int j = ty; j < 32; j += 8) {
            const long long c = c0 + j, r = r0 + tx;
            if (r < rows && c < cols) Elem<T>::st(dst_t + c * rows + r, tile[tx][j]);
        }
    }
}

// flat fp32 -> bf16 image of the whole parameter buffer: 8 elements per thread, 16-byte stores
__global__ __launch_bounds__(256) void cast_flat_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                             long long n8) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256)
        dsvg_pack::cast8(i, src, dst);
}

extern "C" int dsvg_cast_weights(int32_t dtype, const float* src, void* dst, void* dst_t, int64_t rows, int64_t cols,
                                 void* stream) {
    DSVG_CHECK_ARG(src && (dst || dst_t) && rows > 0 && cols > 0, "cast_weights: bad args");
    hipStream_t st = (hipStream_t)stream;
    const long long n = rows * cols;
    if (dtype == DSVG_BF16 && !dst_t && !(n & 7) && !((uintptr_t)src & 15) && !((uintptr_t)dst & 15)) {
        const long long n8 = n / 8;
        const int nb = (int)min((long long)dsvg_cdiv(n8, 256), 2048LL);
        hipLaunchKernelGGL(cast_flat_bf16_kernel, dim3(nb), dim3(256), 0, st, src, (bf16_t*)dst, n8);
        DSVG_LAUNCH_CHECK("cast_weights(flat)");
        return 0;
    }
    dim3 grid(dsvg_cdiv(cols, 32), dsvg_cdiv(rows, 32));
    if (dtype == DSVG_F32)
        hipLaunchKernelGGL(cast_weights_kernel<float>, grid, dim3(256), 0, st, src, (float*)dst, (float*)dst_t,
                           (long long)rows, (long long)cols);
    else if (dtype == DSVG_BF16)
        hipLaunchKernelGGL(cast_weights_kernel<bf16_t>, grid, dim3(256), 0, st, src, (bf16_t*)dst, (bf16_t*)dst_t,
                           (long long)rows, (long long)cols);
    else { dsvg_set_error("cast_weights: bad dtype"); return -1; }
    DSVG_LAUNCH_CHECK("cast_weights");
    return 0;
}

__global__ void advance_step_kernel(long long* counter, uint64_t* seed) { dsvg_pack::advance(counter, seed); }
extern "C" int dsvg_advance_step(int64_t* counter, uint64_t* seed, void* stream) {
    hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (long long*)counter, seed);
    DSVG_LAUNCH_CHECK("advance_step");
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------------
// Up to 32 device-to-device copies in ONE launch: the descriptor table travels in the kernel arguments (hipGraph-capturable,
// no staging buffer).  A hipGraph step reads its inputs and its layout plan from static tensors; refreshing them took ~20
// copy launches of 2-6 us each on the main stream between two replays.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int COPY_MAX = 32;
constexpr int COPY_CHUNK = 256 * 16 * 4;        // bytes per workgroup: 256 threads x 4 pieces of 16 bytes
struct CopyTable {
    const char* src[COPY_MAX];
    char* dst[COPY_MAX];
    long long bytes[COPY_MAX];
    int first_block[COPY_MAX + 1];
    int n;
};
static_assert(sizeof(CopyTable) <= 3600, "the copy table must fit the kernel-argument segment");

__global__ __launch_bounds__(256) void copy_many_kernel(const CopyTable t) {
    int s = 0;
    while (s + 1 < t.n && (int)blockIdx.x >= t.first_block[s + 1]) ++s;
    const long long off = (long long)((int)blockIdx.x - t.first_block[s]) * COPY_CHUNK;
    const long long left = t.bytes[s] - off;
    const char* src = t.src[s] + off;
    char* dst = t.dst[s] + off;
    const long long n = left < COPY_CHUNK ? left : COPY_CHUNK;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        const long long n16 = n >> 4;
        for (long long i = threadIdx.x; i < n16; i += 256)
            reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
        for (long long i = (n16 << 4) + threadIdx.x; i < n; i += 256) dst[i] = src[i];
    } else {
        for (long long i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
    }
}
}  // namespace

extern "C" int dsvg_copy_many(const void* const* src, void* const* dst, const int64_t* bytes, int32_t n, void* stream) {
    DSVG_CHECK_ARG(n >= 0 && (n == 0 || (src && dst && bytes)), "copy_many: bad arguments");
    int at = 0;
    while (at < n) {
        CopyTable t;
        int blocks = 0, k = 0;
        for (; k < COPY_MAX && at < n; ++at) {
            if (bytes[at] <= 0) continue;
            DSVG_CHECK_ARG(src[at] && dst[at], "copy_many: null pointer in entry %d", at);
            t.src[k] = (const char*)src[at];
            t.dst[k] = (char*)dst[at];
            t.bytes[k] = bytes[at];
            t.first_block[k] = blocks;
            blocks += (int)((bytes[at] + COPY_CHUNK - 1) / COPY_CHUNK);
            ++k;
        }
        t.n = k;
        t.first_block[k] = blocks;
        if (k > 0) hipLaunchKernelGGL(copy_many_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t);
    }
    DSVG_LAUNCH_CHECK("copy_many");
    return 0;
}

template <typename T>
__global__ void gate_mul_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ out, long long n,
                                float scale) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        Elem<T>::st(out + i, Elem<T>::ld(y + i) > 0.f ? Elem<T>::ld(dy + i) * scale : 0.f);
}
extern "C" int dsvg_gate_mul(int32_t dtype, const void* dy, const void* y, void* out, int64_t n, float scale,
                             void* stream) {
    DSVG_CHECK_ARG(dy && y && out && n > 0, "gate_mul: bad args");
    const int nb = (int)min(4096LL, (long long)dsvg_cdiv(n, 256));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DSVG_F32)
        hipLaunchKernelGGL(gate_mul_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)dy, (const float*)y,
                           (float*)out, (long long)n, scale);
    else if (dtype == DSVG_BF16)
        hipLaunchKernelGGL(gate_mul_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)y,
                           (bf16_t*)out, (long long)n, scale);
    else { dsvg_set_error("gate_mul: bad dtype"); return -1; }
    DSVG_LAUNCH_CHECK("gate_mul");
    return 0;
}

template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        Elem<T>::st(out + i, Elem<T>::ld(a + i) + Elem<T>::ld(b + i));
}
extern "C" int dsvg_add(int32_t dtype, const void* a, const void* b, void* out, int64_t n, void* stream) {
    DSVG_CHECK_ARG(a && b && out && n > 0, "add: bad args");
    const int nb = (int)min(4096LL, (long long)dsvg_cdiv(n, 256));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DSVG_F32)
        hipLaunchKernelGGL(add_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)out,
                           (long long)n);
    else if (dtype == DSVG_BF16)
        hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b,
                           (bf16_t*)out, (long long)n);
    else { dsvg_set_error("add: bad dtype"); return -1; }
    DSVG_LAUNCH_CHECK("add");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
#include <stdarg.h>
static thread_local char g_err[512] = "";
void dsvg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* dsvg_last_error(void) { return g_err; }
extern "C" int dsvg_version(void) { return DSVG_ABI_VERSION; }
