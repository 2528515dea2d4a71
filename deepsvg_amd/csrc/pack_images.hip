// dsvg_pack_images: every per-step weight image of a bf16 model in ONE launch - the bf16 copy of the flat parameter buffer,
// the fused FFN's forward / backward chunk images + folded bias + fragment-ordered linear2.weight, the fused attention
// image, the attention backward's out_proj^T image, the group-stage layer images - and the step counter / dropout seed
// advance.  A training step replays these 8 launches of 5-15 us each at its start (profiles/r05_graph_step_timeline.csv,
// launches 12-19: 60 us); they only read the flat fp32 parameters, so they are one grid cut into block ranges.
// The bodies are pack_images.h's, shared with the stand-alone launches, which stay (tests compare the two bit for bit).
#include "pack_images.h"
#include "../../include/dsvg.h"

namespace {

constexpr int CAST_U = 4;       // 16-byte groups of the flat copy per thread

struct PackTable {
    const float* flat;
    bf16_t* flat_lp;
    long long n8;
    const int64_t* ffn_offs;
    bf16_t* ffn_fwd;
    bf16_t* ffn_bwd;
    float* ffn_b1f;
    bf16_t* ffn_w2p;
    const int64_t* attn_offs;
    bf16_t* attn_img;
    bf16_t* attn_bwd;
    const int64_t* gs_offs;
    bf16_t* gs_fwd;
    bf16_t* gs_bwd;
    long long* counter;
    uint64_t* seed;
    int ffn_layers, attn_layers, gs_layers;
    int first[9];               // first block of segment k (first[8] = grid size)
};

__global__ __launch_bounds__(256) void pack_images_kernel(const PackTable t) {
    const int b = blockIdx.x;
    int k = 0;
    while (k < 7 && b >= t.first[k + 1]) ++k;           // (uniform over the workgroup)
    const long long vb = b - t.first[k];
    const long long gid = vb * 256 + threadIdx.x;
    switch (k) {
    case 0:
#pragma unroll
        for (int u = 0; u < CAST_U; ++u) {
            const long long i = (vb * CAST_U + u) * 256 + threadIdx.x;
            if (i < t.n8) dsvg_pack::cast8(i, t.flat, t.flat_lp);
        }
        break;
    case 1: dsvg_pack::ffn_slot(gid, t.flat, t.ffn_offs, t.ffn_layers, t.ffn_fwd, t.ffn_bwd); break;
    case 2: dsvg_pack::ffn_w2p_slot(gid, t.flat, t.ffn_offs, t.ffn_layers, t.ffn_w2p); break;
    case 3: dsvg_pack::ffn_fold_bias_row((int)vb * 4 + (threadIdx.x >> 6), threadIdx.x & 63, t.flat, t.ffn_offs, t.ffn_layers, t.ffn_b1f); break;
    case 4: dsvg_pack::attn_slot(gid, t.flat, t.attn_offs, t.attn_layers, t.attn_img); break;
    case 5: dsvg_pack::attn_bwd_slot(gid, t.flat, t.attn_offs, t.attn_layers, t.attn_bwd); break;
    case 6: dsvg_pack::gs_slot(gid, t.flat, t.gs_offs, t.gs_layers, t.gs_fwd, t.gs_bwd); break;
    default:
        if (threadIdx.x == 0) dsvg_pack::advance(t.counter, t.seed);
    }
}

}  // namespace

extern "C" int dsvg_pack_images(const float* flat_f32, void* flat_bf16, int64_t n,
                                const int64_t* ffn_offs, int32_t ffn_layers, void* ffn_fwd, void* ffn_bwd, float* ffn_b1f,
                                void* ffn_w2p,
                                const int64_t* attn_offs, int32_t attn_layers, void* attn_img, void* attn_bwd,
                                const int64_t* gs_offs, int32_t gs_layers, void* gs_fwd, void* gs_bwd,
                                int64_t* counter, uint64_t* seed, void* stream) {
    DSVG_CHECK_ARG(flat_f32 && flat_bf16 && n > 0 && !(n & 7) && !((uintptr_t)flat_f32 & 15) && !((uintptr_t)flat_bf16 & 15),
                   "pack_images: the flat buffers must be 16-byte aligned and hold a multiple of 8 elements");
    DSVG_CHECK_ARG(ffn_layers >= 0 && attn_layers >= 0 && gs_layers >= 0, "pack_images: bad layer counts");
    DSVG_CHECK_ARG(!ffn_layers || (ffn_offs && ffn_fwd && ffn_bwd && ffn_b1f), "pack_images: FFN images missing");
    DSVG_CHECK_ARG(!attn_layers || (attn_offs && attn_img), "pack_images: attention image missing");
    DSVG_CHECK_ARG(!gs_layers || (gs_offs && gs_fwd && gs_bwd), "pack_images: group-stage images missing");
    PackTable t;
    t.flat = flat_f32; t.flat_lp = (bf16_t*)flat_bf16; t.n8 = n / 8;
    t.ffn_offs = ffn_offs; t.ffn_fwd = (bf16_t*)ffn_fwd; t.ffn_bwd = (bf16_t*)ffn_bwd; t.ffn_b1f = ffn_b1f; t.ffn_w2p = (bf16_t*)ffn_w2p;
    t.attn_offs = attn_offs; t.attn_img = (bf16_t*)attn_img; t.attn_bwd = (bf16_t*)attn_bwd;
    t.gs_offs = gs_offs; t.gs_fwd = (bf16_t*)gs_fwd; t.gs_bwd = (bf16_t*)gs_bwd;
    t.counter = (long long*)counter; t.seed = seed;
    t.ffn_layers = ffn_layers; t.attn_layers = attn_layers; t.gs_layers = gs_layers;
    const long long blocks[8] = {
        dsvg_cdiv(t.n8, 256LL * CAST_U),
        dsvg_cdiv((long long)ffn_layers * dsvg_pack::FFN_SLOTS, 256LL),
        ffn_w2p ? dsvg_cdiv((long long)ffn_layers * (dsvg_pack::D * dsvg_pack::F / 8), 256LL) : 0,
        dsvg_cdiv((long long)ffn_layers * dsvg_pack::F, 4LL),
        dsvg_cdiv((long long)attn_layers * dsvg_pack::ATTN_IMG_FRAGS * 64, 256LL),
        attn_bwd ? dsvg_cdiv((long long)attn_layers * dsvg_pack::ATTN_BWD_SLOTS, 256LL) : 0,
        dsvg_cdiv((long long)gs_layers * dsvg_pack::GS_SLOTS, 256LL),
        (counter || seed) ? 1 : 0,
    };
    long long at = 0;
    for (int k = 0; k < 8; ++k) { t.first[k] = (int)at; at += blocks[k]; }
    t.first[8] = (int)at;
    DSVG_CHECK_ARG(at < (1LL << 31), "pack_images: too many blocks");
    hipLaunchKernelGGL(pack_images_kernel, dim3((unsigned)at), dim3(256), 0, (hipStream_t)stream, t);
    DSVG_LAUNCH_CHECK("pack_images");
    return 0;
}
