// Device-side batch assembly: the per-item cat / pad chain of SVGTensorDataset.get_data
// (deepsvg/svgtensor_dataset.py:164-205; SVGTensor.add_eos/add_sos/pad, deepsvg/difflib/tensor.py:108-143) and the
// DataLoader's default collate, as one gather over a packed icon store.
//
// Store layout (built once on the host, deepsvg_amd/dataset.py):
//   rows     int16 [R, 12]   one drawing command per row: (command, 11 arguments in SVGTensor.arg_keys order);
//                            the values are the numericalised integers -1..255 of the .pkl tensors
//   slot_off int32 [n_slots+1]  row range of slot (variant, group) = variant*G + group, a variant being one stored
//                               (icon, augmentation) pair; the groups of a variant are stored back to back, so the
//                               "grouped" sequence (all groups concatenated, svgtensor_dataset.py:175) is the span
//                               of its G slots
// Pure integer / byte movement: HBM-bound, 24 B read + 48..92 B written per output token; one thread per token,
// consecutive threads write consecutive tokens.
#include "dsvg_common.h"
#include "../../include/dsvg.h"

namespace {
constexpr int N_ARGS = 11;
constexpr int ROW_W = 12;          // int16 per stored row
constexpr int CMD_EOS = 4, CMD_SOS = 5, N_CMD = 7;

// deepsvg/difflib/tensor.py:15-21, bit a of entry c = CMD_ARGS_MASK[c][a]
__device__ __constant__ uint32_t kCmdArgsMask[N_CMD] = {0x600u, 0x600u, 0x7E0u, 0x61Fu, 0u, 0u, 0u};

struct Row {
    float cmd;
    float a[N_ARGS];
};

__device__ __forceinline__ Row load_row(const int16_t* __restrict__ rows, long long r) {
    // 24-byte rows: three 8-byte loads (the array base is at least 8-byte aligned)
    const uint2* p = reinterpret_cast<const uint2*>(rows + r * ROW_W);
    const uint2 q0 = p[0], q1 = p[1], q2 = p[2];
    const uint32_t w[6] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y};
    Row o;
    o.cmd = (float)(int16_t)(w[0] & 0xFFFFu);
#pragma unroll
    for (int a = 0; a < N_ARGS; ++a) {
        const int e = a + 1;
        o.a[a] = (float)(int16_t)((w[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu);
    }
    return o;
}

__global__ void assemble_kernel(const int16_t* __restrict__ rows, long long n_rows,
                                const int32_t* __restrict__ slot_off, long long n_slots,
                                const int32_t* __restrict__ variant, int G, int grouped, long long n_seq, int L,
                                float pad_val, float rel_shift,
                                float* __restrict__ commands, float* __restrict__ args,
                                float* __restrict__ args_rel) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_seq * L) return;
    const long long q = t / L;
    const int s = (int)(t % L);
    const long long n = grouped ? q : q / G;
    const int g = grouped ? 0 : (int)(q % G);
    const int span = grouped ? G : 1;
    long long slot = (long long)variant[n] * G + g;
    slot = min(max(slot, 0ll), n_slots - span);               // never read outside the store
    const long long begin = slot_off[slot];
    const int len = min((int)(slot_off[slot + span] - begin), L - 2);

    Row o;
    o.cmd = (s == 0) ? (float)CMD_SOS : (float)CMD_EOS;          // SOS, then EOS as end marker and as padding
#pragma unroll
    for (int a = 0; a < N_ARGS; ++a) o.a[a] = pad_val;
    const bool data = s >= 1 && s - 1 < len;
    const long long r = min(begin + max(s - 1, 0), n_rows - 1);
    const Row ld = load_row(rows, r);          // unconditional load from a clamped address, selected afterwards
    if (data) o = ld;

    commands[t] = o.cmd;
    if (args) {
#pragma unroll
        for (int a = 0; a < N_ARGS; ++a) args[t * N_ARGS + a] = o.a[a];
    }
    if (args_rel) {
        // SVGTensor.get_relative_args (tensor.py:172-189): positions of every real command (m/l/c/a) but the first
        // are taken relative to the end position of the previous real command; used slots are shifted by ARGS_DIM-1,
        // unused ones are PAD_VAL
        const int c = (int)o.cmd;
        float v[N_ARGS];
#pragma unroll
        for (int a = 0; a < N_ARGS; ++a) v[a] = o.a[a];
        if (data && c < CMD_EOS) {
            long long j = r - 1;
            bool found = false;
            float px = 0.f, py = 0.f;
            while (j >= begin) {                                 // previous real command (normally row r-1)
                const int16_t* pr = rows + j * ROW_W;
                if (pr[0] < CMD_EOS) {
                    px = (float)pr[10];
                    py = (float)pr[11];
                    found = true;
                    break;
                }
                --j;
            }
            if (found) {
                v[5] -= px; v[6] -= py; v[7] -= px; v[8] -= py; v[9] -= px; v[10] -= py;
            }
        }
        const uint32_t m = (c >= 0 && c < N_CMD) ? kCmdArgsMask[c] : 0u;
#pragma unroll
        for (int a = 0; a < N_ARGS; ++a) args_rel[t * N_ARGS + a] = ((m >> a) & 1u) ? v[a] + rel_shift : pad_val;
    }
}
}  // namespace

extern "C" int dsvg_assemble_batch(const int16_t* rows, int64_t n_rows, const int32_t* slot_off, int64_t n_slots,
                                   const int32_t* variant, int64_t n_items, int32_t G, int32_t grouped,
                                   int32_t L, float pad_val, int32_t args_dim, float* commands, float* args,
                                   float* args_rel, void* stream) {
    DSVG_CHECK_ARG(rows && slot_off && variant && commands, "assemble_batch: null pointer");
    DSVG_CHECK_ARG(n_rows > 0 && n_items > 0 && G > 0 && L >= 2 && n_slots >= G && n_slots % G == 0,
                   "assemble_batch: bad shape (n_rows=%lld n_items=%lld G=%d L=%d n_slots=%lld)", (long long)n_rows,
                   (long long)n_items, G, L, (long long)n_slots);
    DSVG_CHECK_ARG(((uintptr_t)rows & 7) == 0, "assemble_batch: rows must be 8-byte aligned");
    const long long n_seq = grouped ? n_items : n_items * G;
    DSVG_CHECK_ARG(n_seq * L < (1ll << 40), "assemble_batch: batch too large");
    hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)dsvg_cdiv(n_seq * L, 256)), dim3(256), 0, (hipStream_t)stream,
                       rows, (long long)n_rows, slot_off, (long long)n_slots, variant, G, grouped, n_seq, L,
                       pad_val, (float)(args_dim - 1), commands, args, args_rel);
    DSVG_LAUNCH_CHECK("assemble_batch");
    return 0;
}
