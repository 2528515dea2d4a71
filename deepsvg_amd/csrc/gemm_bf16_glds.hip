// bf16 GEMM (fp32 accumulate) for the aligned shapes of the hot path: tokens x {256..768} x {256,512} forward /
// input-gradient GEMMs and the split-K weight-gradient GEMM.  Same math and options as gemm_bf16.hip, built around
// what bounds these short-K GEMMs on gfx950 (memory latency per workgroup and LDS cycles, not MFMA issue):
//   * operand tiles go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass, so the
//     kernel fits 4 workgroups per CU and the whole K step of a tile is in flight at once.  The DMA writes lane-linear
//     (wave-uniform base + 16 B x lane), so the bank swizzle is applied to the per-lane SOURCE address:
//        k-contiguous operand  image [128 mn][64 k] (128 B rows): 16-byte chunk c of row r lives at c ^ ((r>>1)&7)
//        mn-contiguous operand image [64 k][128 mn] (256 B rows): chunk c of k-row k lives at c ^ ((k&3)<<2)
//     both conflict-free for ds_read_b128 / ds_read_b64_tr_b16 fragment reads (lane groups of MI355X_MICROARCH §LDS).
//   * the MFMA is issued with the operands swapped (D = W_frag x X_frag, i.e. the transposed tile): a lane then owns
//     one token row and 4 consecutive output columns per accumulator quad; two v_permlane32_swap rounds turn that
//     into 16 consecutive columns, so bias / ReLU / dropout / residual / gate and the stores (32 contiguous bytes
//     per lane, 64 per lane pair, a row's four pieces issued back to back) run from registers - the epilogue uses
//     no LDS at all (the register-staged kernel spends ~1/3 of its LDS cycles there).
//   * out-of-range rows are clamped on the load side (their results are never stored): no exec-mask branches.
// Eligibility is decided on the host (dsvg_gemm_bf16_glds_try); everything else runs on gemm_bf16.hip.
#include "gemm_bf16.h"
#include <map>
#include <mutex>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short shortx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int GBM = 128, GBN = 128, GBK = 64;
constexpr int IMG = 128 * 64;          // elements per operand image (16 KiB)

#define DSVG_LDS_PTR(p) ((void __attribute__((address_space(3)))*)(p))
#define DSVG_GLB_PTR(p) ((const void __attribute__((address_space(1)))*)(p))

union Frag8 {
    bf16x8 v;
    shortx4 h[2];
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

// One K step of both operands by LDS-DMA issued from inline asm (8 instructions per wave: 4 pieces of A, 4 of B), for the
// 4-stage variant: hipcc drains vmcnt(0) before the first ds_read after a *builtin* LDS-DMA, which would serialise a deep
// prefetch; from asm the waits are ours (counted s_waitcnt vmcnt, only loads in flight inside the K loop).
// a0..a3 / b0..b3: per-lane global addresses of the pieces; lds_a / lds_b: wave-uniform LDS byte addresses of piece 0.
__device__ __forceinline__ void dma_step8(const char* a0, const char* a1, const char* a2, const char* a3, const char* b0,
                                          const char* b1, const char* b2, const char* b3, uint32_t lds_a, uint32_t lds_b) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %9\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, off\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, off\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, off\n\t"
        "s_mov_b32 m0, %10\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %5, off\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %6, off\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %7, off\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %8, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "s"(lds_a), "s"(lds_b)
        : "memory", "scc");
}

// NST: LDS stages (1: 32 KiB, 2: 64 KiB; 4: 128 KiB of dynamic LDS, asm DMA three K steps ahead - for launches of at most
// one workgroup per CU, where nothing else on the CU hides the ~1.4 us of a DMA round trip per K step); OCC: waves per SIMD the register allocation is held to (4 -> 128 VGPRs, no
// spills; 5 -> 96 VGPRs, a few epilogue values spill, but all five 32-KiB workgroups a CU's LDS can hold are resident:
// the packed / live-prefix GEMMs launch ~1.25 x 1024 workgroups, which then run as one wave of workgroups, not two)
// the whole kernel as a function of (block index x, block index y): the plain kernel passes blockIdx, the grouped
// weight-gradient kernel its position inside the problem it belongs to
template <bool AKC, bool BKC, int EPI, int NST>
__device__ __forceinline__ void glds_body(const dsvg_gemm_desc& p, int tiles_n, int nwg_mn, int k_chunk, float* part,
                                          float* rs_part, int mode, int part_bf16, int bid, int bid_y) {
    __shared__ __attribute__((aligned(1024))) bf16_t smem_static[NST <= 2 ? NST * 2 * IMG : 8];
    extern __shared__ __attribute__((aligned(1024))) bf16_t smem_dynamic[];
    bf16_t* const smem = NST <= 2 ? smem_static : smem_dynamic;

    // tile schedule: identical to gemm_bf16.hip modes 0 and 1 (workgroup b runs on XCD b % 8)
    const int xcd = bid % 8, local = bid / 8;
    int wgid, kz;
    if (mode == 1) {
        wgid = local % nwg_mn;
        kz = (local / nwg_mn) * 8 + xcd;
    } else {
        const int q8 = nwg_mn / 8, r8 = nwg_mn % 8;
        wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
        kz = bid_y;
    }
    const int tile_m = wgid / tiles_n, tile_n = wgid % tiles_n;
    const int m0 = tile_m * GBM, n0 = tile_n * GBN;
    const int k_begin = kz * k_chunk;
    const int k_end = min(p.K, k_begin + k_chunk);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5;

    // ---- LDS-DMA source offsets (bytes, relative to the operand's address at the current k0) -----------------
    uint32_t offa[4], offb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wave * 4 + i;                 // 1 KiB piece of the 16 KiB image this instruction fills
        if (AKC) {
            const int r = piece * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            offa[i] = (uint32_t)(((size_t)min(m0 + r, p.M - 1) * p.lda + 8 * c) * 2);
        } else {
            const int kr = piece * 4 + (lane >> 4);
            const int c = (lane & 15) ^ ((lane >> 4) << 2);
            offa[i] = (uint32_t)(((size_t)kr * p.lda + min(m0 + 8 * c, ((p.M + 7) & ~7) - 8)) * 2);
        }
        if (BKC) {
            const int r = piece * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            offb[i] = (uint32_t)(((size_t)min(n0 + r, p.N - 1) * p.ldb + 8 * c) * 2);
        } else {
            const int kr = piece * 4 + (lane >> 4);
            const int c = (lane & 15) ^ ((lane >> 4) << 2);
            offb[i] = (uint32_t)(((size_t)kr * p.ldb + min(n0 + 8 * c, p.N - 8)) * 2);
        }
    }
    auto stage = [&](int k0, int buf) {
        const char* ab = (const char*)p.A + (AKC ? (size_t)k0 * 2 : (size_t)k0 * p.lda * 2);
        const char* bb = (const char*)p.B + (BKC ? (size_t)k0 * 2 : (size_t)k0 * p.ldb * 2);
        bf16_t* da = smem + buf * 2 * IMG + wave * 2048;
        bf16_t* db = da + IMG;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds(DSVG_GLB_PTR(ab + offa[i]), DSVG_LDS_PTR(da + i * 512), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(DSVG_GLB_PTR(bb + offb[i]), DSVG_LDS_PTR(db + i * 512), 16, 0, 0);
        }
    };

    // ---- fragment read offsets (bytes inside one operand image) ------------------------------------------------
    // k-contiguous image: row r = base + (lane & 31), chunk 2 kk + h, swizzle (r >> 1) & 7 = (lane >> 1) & 7
    uint32_t fk[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fk[kk] = (uint32_t)(((lane & 31) * 64 + 8 * ((2 * kk + h) ^ ((lane >> 1) & 7))) * 2);
    // mn-contiguous image (hardware-transpose read): logical (k row, column) of gemm_bf16.hip's frag_tr, physical
    // chunk = (column >> 3) ^ ((k row & 3) << 2)
    const int g = lane >> 4, q = lane & 15;
    auto tr_off = [&](int w, int it) -> uint32_t {
        const int krow = 8 * (g >> 1) + (q >> 2);
        const int chunk = (8 * w + 4 * it + 2 * (g & 1) + ((q & 3) >> 1)) ^ (((q >> 2) & 3) << 2);
        return (uint32_t)((krow * 128 + 8 * chunk + 4 * (q & 1)) * 2);
    };
    const uint32_t ta0 = tr_off(wm, 0), ta1 = tr_off(wm, 1), tb0 = tr_off(wn, 0), tb1 = tr_off(wn, 1);

    auto frag_kc = [&](const char* img, int w, int it, int kk) -> bf16x8 {
        Frag8 f;
        f.u = *reinterpret_cast<const uint4*>(img + (w * 64 + it * 32) * 128 + fk[kk]);
        return f.v;
    };
    auto frag_tr = [&](const char* img, uint32_t off, int kk) -> bf16x8 {
        Frag8 f;
        f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((shortx4 __attribute__((address_space(3)))*)(img + off + kk * 4096));
        f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (shortx4 __attribute__((address_space(3)))*)(img + off + kk * 4096 + 1024));
        return f.v;
    };

    // Round 6: the weight operand B of a forward / input-gradient GEMM (<= 1.4 MB, read by every workgroup) is cold in this
    // XCD's L2 when the launch starts - inside a training step every layer's weights are touched a few times per step - and
    // the workgroups walk it in lockstep, K step by K step: each step would begin with an HBM miss they all wait for.  The
    // workgroups of an XCD (workgroup b runs on XCD b % 8) request all of it up front, 4 KiB each, by LDS-DMA into this wave's
    // own first piece of stage 0 (its own first real DMA overwrites it, in order; nothing reads it before).
    if (EPI != EPI_PARTIAL && (part_bf16 & 16)) {        // (bit 4 of this launch argument: dsvg_gemm_bf16_glds launch())
        const size_t span = (BKC ? (size_t)p.N * p.ldb : (size_t)p.K * p.ldb) * 2;
        const unsigned n_pc = (unsigned)((span + 4095) >> 12);
        const size_t off = (size_t)(((unsigned)bid >> 3) % n_pc) * 4096 + (size_t)wave * 1024 + (size_t)lane * 16;
        const char* src = (const char*)p.B + (off + 16 <= span ? off : span - 16);
        const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)DSVG_LDS_PTR(smem) + (uint32_t)wave * 4096u);
        uint32_t keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, off\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }

    // acc[jn][im]: transposed 32x32 tiles, D[i = output column][j = token row]
    floatx16 acc00, acc01, acc10, acc11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }
    // row sums of the token-side operand (bias gradient inside the weight-gradient GEMM): the lane's 8 k values of
    // token row (lane & 31) are summed with v_dot2c_f32_bf16 against packed ones - 2 VGPRs instead of the 32 an
    // all-ones MFMA would need; only the column-tile-0 workgroups do it, and the two waves that hold the same token
    // rows (wn = 0, 1) split the K step between them (the extra VALU work sits on those waves' critical path)
    const bool do_rs = (EPI == EPI_PARTIAL) && rs_part != nullptr && tile_n == 0;
    float rs0 = 0.f, rs1 = 0.f;
    auto rowsum8 = [](const bf16x8& f, float s) -> float {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        Frag8 t;
        t.v = f;
        const uint32_t w[4] = {t.u.x, t.u.y, t.u.z, t.u.w};
        const bf16x2 one = __builtin_bit_cast(bf16x2, 0x3f803f80u);
#pragma unroll
        for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w[e]), one, s, false);
        return s;
    };

    auto compute = [&](int buf) {
        const char* ia = (const char*)(smem + buf * 2 * IMG);
        const char* ib = ia + IMG * 2;
#pragma unroll
        for (int kk = 0; kk < GBK / 16; ++kk) {
            bf16x8 a0, a1, b0, b1;
            if (AKC) { a0 = frag_kc(ia, wm, 0, kk); a1 = frag_kc(ia, wm, 1, kk); }
            else     { a0 = frag_tr(ia, ta0, kk);   a1 = frag_tr(ia, ta1, kk); }
            if (BKC) { b0 = frag_kc(ib, wn, 0, kk); b1 = frag_kc(ib, wn, 1, kk); }
            else     { b0 = frag_tr(ib, tb0, kk);   b1 = frag_tr(ib, tb1, kk); }
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, acc11, 0, 0, 0);
            if (EPI == EPI_PARTIAL && do_rs && (kk >> 1) == wn) { rs0 = rowsum8(a0, rs0); rs1 = rowsum8(a1, rs1); }
        }
    };

    if (NST == 4) {
        // stage s lives in slot s % 4; stages s + 1 .. s + 3 are in flight while stage s is consumed
        const uint32_t lds0 = (uint32_t)(uintptr_t)DSVG_LDS_PTR(smem) + wave * 4096;
        auto issue = [&](int s) {
            const int k0 = k_begin + s * GBK;
            const char* ab = (const char*)p.A + (AKC ? (size_t)k0 * 2 : (size_t)k0 * p.lda * 2);
            const char* bb = (const char*)p.B + (BKC ? (size_t)k0 * 2 : (size_t)k0 * p.ldb * 2);
            const uint32_t la = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(s & 3) * (2 * IMG * 2));
            dma_step8(ab + offa[0], ab + offa[1], ab + offa[2], ab + offa[3], bb + offb[0], bb + offb[1], bb + offb[2],
                      bb + offb[3], la, la + IMG * 2);
        };
        const int n_steps = (k_end - k_begin + GBK - 1) / GBK;
        for (int s = 0; s < 3 && s < n_steps; ++s) issue(s);
        for (int s = 0; s < n_steps; ++s) {
            const int ahead = min(n_steps - 1, s + 2) - s;      // stages issued behind stage s (8 DMA instructions each)
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();       // stage s landed everywhere; everybody is done with stage s - 1's slot
            if (s + 3 < n_steps) issue(s + 3);
            compute(s & 3);
        }
    } else if (NST == 1) {
        for (int k0 = k_begin; k0 < k_end; k0 += GBK) {
            stage(k0, 0);
            __syncthreads();            // every wave drains its DMA (vmcnt 0) before the barrier
            compute(0);
            __syncthreads();            // all fragment reads done before the image is overwritten
        }
    } else {
        if (k_begin < k_end) stage(k_begin, 0);
        int buf = 0;
        for (int k0 = k_begin; k0 < k_end; k0 += GBK) {
            __syncthreads();            // tile k0 landed everywhere; everybody is done reading the other buffer
            if (k0 + GBK < k_end) stage(k0 + GBK, buf ^ 1);
            compute(buf);
            buf ^= 1;
        }
    }

    const int mrow = m0 + wm * 64 + (lane & 31);        // + 32 im
    const int ncol = n0 + wn * 64;                       // + 32 jn + ...

    if (EPI == EPI_PARTIAL) {       // split-K slice: raw fp32 accumulators, 16-byte stores (4 consecutive columns)
        const size_t slice = dsvg_splitk_slice(p.M, p.N, rs_part != nullptr);
        float* my_part = part + (size_t)kz * slice;
        auto put = [&](const floatx16& c, int jn, int im) {
            const int m = mrow + 32 * im;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = ncol + 32 * jn + 8 * gq + 4 * h;
                if (m < p.M && n < p.N)
                    *reinterpret_cast<float4*>(my_part + (size_t)m * p.N + n) =
                        make_float4(c[4 * gq], c[4 * gq + 1], c[4 * gq + 2], c[4 * gq + 3]);
            }
        };
        // bf16 slices (default for the bf16 weight-gradient GEMMs, N % 8 == 0): the partials are 40 % of such a GEMM's HBM
        // traffic and everything of the reduction's; a slice sums ~1-2 k tokens, 32-64 slices are added in fp32 in a fixed
        // order, so the rounding (2^-9 relative per slice, independent) stays far below the bf16 operand rounding.  One
        // v_permlane32_swap pair per two quads gives each lane 8 consecutive columns: one 16-byte store.
        auto put_bf16 = [&](const floatx16& c, int jn, int im) {
            const int m = mrow + 32 * im;
            bf16_t* base = reinterpret_cast<bf16_t*>(my_part);
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                const uint32_t a0 = f2bf_pk(c[8 * gp + 0], c[8 * gp + 1]), a1 = f2bf_pk(c[8 * gp + 2], c[8 * gp + 3]);
                const uint32_t b0 = f2bf_pk(c[8 * gp + 4], c[8 * gp + 5]), b1 = f2bf_pk(c[8 * gp + 6], c[8 * gp + 7]);
                auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                const int n = ncol + 32 * jn + 16 * gp + 8 * h;
                if (m < p.M && n < p.N)
                    *reinterpret_cast<uint4*>(base + (size_t)m * p.N + n) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            }
        };
        if (part_bf16 & 1) { put_bf16(acc00, 0, 0); put_bf16(acc01, 0, 1); put_bf16(acc10, 1, 0); put_bf16(acc11, 1, 1); }
        else { put(acc00, 0, 0); put(acc01, 0, 1); put(acc10, 1, 0); put(acc11, 1, 1); }
        if (do_rs) {                // lanes l and l+32 hold the two k halves of token row (lane & 31)
            rs0 += __shfl_xor(rs0, 32, 64);
            rs1 += __shfl_xor(rs1, 32, 64);
            // the wn = 1 wave hands its half of every K step over through LDS (the operand images are dead by now;
            // do_rs is uniform over the workgroup, so the barrier is legal); fixed order: wn 0 + wn 1
            float* rsx = reinterpret_cast<float*>(smem);
            __syncthreads();        // every wave is done reading the operand images (2-stage loop has no trailing barrier)
            if (wn == 1 && h == 0) { rsx[wm * 64 + (lane & 31)] = rs0; rsx[wm * 64 + 32 + (lane & 31)] = rs1; }
            __syncthreads();
            if (wn == 0 && h == 0) {
                rs0 += rsx[wm * 64 + (lane & 31)];
                rs1 += rsx[wm * 64 + 32 + (lane & 31)];
                if (mrow < p.M) rs_part[(size_t)kz * slice + mrow] = rs0;
                if (mrow + 32 < p.M) rs_part[(size_t)kz * slice + mrow + 32] = rs1;
            }
        }
        return;
    }

    // ---- register epilogue ------------------------------------------------------------------------------------
    constexpr bool has_res = EPI == EPI_BIAS_RES_DROP;
    constexpr bool has_gate = EPI == EPI_GATE;
    constexpr bool relu = EPI == EPI_BIAS_RELU_DROP;
    constexpr bool may_drop = EPI == EPI_BIAS_RES_DROP || EPI == EPI_BIAS_RELU_DROP;
    const DropCtx dc = drop_make(may_drop ? p.drop_p : 0.f, p.seed, p.drop_site);
    const bool has_bias = !has_gate && p.bias != nullptr;
    // (a bias that is a row range of a longer vector - the argument head's slots in use - need not start on a 16-byte boundary)
    const bool bias_al = (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0;
    const bool n_aligned = !(p.N & 7);     // else: the last chunk of a row is partial and dropout ids are unaligned

    // One 32x32 accumulator tile -> 16 consecutive output columns of token row m per lane (two permlane32 exchange
    // rounds: quads 8 g + 4 h + e  ->  octets  ->  16-column runs n = 32 jn + 16 h + i), epilogue math on two aligned
    // 8-column chunks, results left packed in `pk` so that the caller can issue all stores of a 128-byte output line
    // back to back (PMC: stores interleaved with the dropout hash reach HBM as partial lines, WRITE_SIZE +38 %; 16-byte
    // accesses covering only 32 contiguous bytes per row also doubled the gate / residual FETCH_SIZE).
    auto tile16 = [&](const floatx16& c, int jn, int im, uint4 (&pk)[2], int (&nvalid)[2]) {
        const int m = mrow + 32 * im;
        uint32_t x[4][4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
#pragma unroll
            for (int e = 0; e < 4; ++e) x[gq][e] = __float_as_uint(c[4 * gq + e]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {       // round 1: (x0, x1) and (x2, x3) -> 8 consecutive columns each
            auto s01 = __builtin_amdgcn_permlane32_swap(x[0][e], x[1][e], false, false);
            auto s23 = __builtin_amdgcn_permlane32_swap(x[2][e], x[3][e], false, false);
            x[0][e] = s01[0]; x[1][e] = s01[1]; x[2][e] = s23[0]; x[3][e] = s23[1];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {       // round 2: low lanes keep columns 0..15, high lanes 16..31
            auto s02 = __builtin_amdgcn_permlane32_swap(x[0][e], x[2][e], false, false);
            auto s13 = __builtin_amdgcn_permlane32_swap(x[1][e], x[3][e], false, false);
            x[0][e] = s02[0]; x[2][e] = s02[1]; x[1][e] = s13[0]; x[3][e] = s13[1];
        }
        const int n16 = ncol + 32 * jn + 16 * h;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(x[2 * cb][e]); v[4 + e] = __uint_as_float(x[2 * cb + 1][e]); }
            const int nb = n16 + 8 * cb;
            const int nv = (m < p.M && nb < p.N) ? min(8, p.N - nb) : 0;   // valid columns of this chunk
            nvalid[cb] = nv;
            if (nv == 0) { pk[cb] = make_uint4(0u, 0u, 0u, 0u); continue; }
            if (has_bias) {
                if (nv == 8 && bias_al) {
                    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + nb);
                    const float4 b1 = *reinterpret_cast<const float4*>(p.bias + nb + 4);
                    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
                    v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += e < nv ? p.bias[nb + e] : 0.f;
                }
            }
            if (relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (has_gate) {
                float gv[8];
                const bf16_t* gp = (const bf16_t*)p.gate + (size_t)m * p.ldgate + nb;
                if (nv == 8) unpack8(*reinterpret_cast<const uint4*>(gp), gv);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) gv[e] = e < nv ? bf2f(gp[e]) : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gv[e] > 0.f ? v[e] * p.gate_scale : 0.f;
            }
            float rv[8];
            if (has_res) {
                const bf16_t* rp = (const bf16_t*)p.res + (size_t)m * p.ldres + nb;
                if (nv == 8) unpack8(*reinterpret_cast<const uint4*>(rp), rv);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) rv[e] = e < nv ? bf2f(rp[e]) : 0.f;
                }
                if (p.res_pre) {        // (run-time: the residual inside the dropout)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rv[e];
                }
            }
            if (may_drop && dc.on) {
                float dm[8];
                if (n_aligned) {
                    drop_mult8(dc, (uint64_t)m * p.N + nb, dm);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) dm[e] = drop_mult(dc, (uint64_t)m * p.N + nb + e);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= dm[e];
            }
            if (has_res && !p.res_pre) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rv[e];
            }
            pk[cb] = pack8(v);
        }
    };
    auto put8 = [&](bf16_t* cp, const uint4& v, int nv) {
        if (nv == 8) { *reinterpret_cast<uint4*>(cp) = v; return; }
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (e < nv) cp[e] = (bf16_t)((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu));
    };
    // the 64 columns of the wave's sub-tile = one 128-byte line per token row: four 32-byte pieces (2 lanes x 2 tiles)
    auto row_block = [&](const floatx16& c0, const floatx16& c1, int im) {
        uint4 pk0[2], pk1[2];
        int nv0[2], nv1[2];
        tile16(c0, 0, im, pk0, nv0);
        tile16(c1, 1, im, pk1, nv1);
        bf16_t* cp = (bf16_t*)p.C + (size_t)(mrow + 32 * im) * p.ldc + ncol + 16 * h;
        put8(cp, pk0[0], nv0[0]);
        put8(cp + 8, pk0[1], nv0[1]);
        put8(cp + 32, pk1[0], nv1[0]);
        put8(cp + 40, pk1[1], nv1[1]);
    };
    row_block(acc00, acc10, 0);
    row_block(acc01, acc11, 1);
}

template <bool AKC, bool BKC, int EPI, int NST, int OCC>
__global__ __launch_bounds__(256, NST == 1 ? OCC : 2) void gemm_bf16_glds_kernel(dsvg_gemm_desc p, int tiles_n, int nwg_mn,
                                                                              int k_chunk, float* part, float* rs_part,
                                                                              int mode, int part_bf16) {
    glds_body<AKC, BKC, EPI, NST>(p, tiles_n, nwg_mn, k_chunk, part, rs_part, mode, part_bf16, (int)blockIdx.x, (int)blockIdx.y);
}

// ---------------------------------------------------------------------------------------------------------------------
// Grouped launch of split-K weight-gradient GEMMs (both operands token-major, 4-stage variant): the dW GEMMs of a layer of
// the 4096-row stages are 5-8 us launches of 64-256 workgroups each, back to back and independent of each other; queued
// under dsvg_gemm_group_scope they run as ONE launch whose workgroups look their problem up in a table that travels in the
// kernel arguments (capturable).  Each problem keeps the grid, slices and workspace it would have had on its own, so the
// results are bit-identical to separate launches.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int GROUP_MAX = 16;
struct WgItem {
    const void* A; const void* B; float* part; float* rs_part;
    int M, N, K, tiles_n, nwg_mn, k_chunk, mode, part_bf16, first_block, grid_x;
    long long lda, ldb;
};
struct WgTable { WgItem it[GROUP_MAX]; int n; };
static_assert(sizeof(WgTable) <= 3600, "the problem table must fit the kernel-argument segment");

__global__ __launch_bounds__(256, 2) void gemm_bf16_wgrad_group_kernel(const WgTable t) {
    int g = 0;
    while (g + 1 < t.n && (int)blockIdx.x >= t.it[g + 1].first_block) ++g;
    const WgItem& w = t.it[g];
    dsvg_gemm_desc p;
    p.A = w.A; p.B = w.B; p.M = w.M; p.N = w.N; p.K = w.K; p.lda = w.lda; p.ldb = w.ldb;
    const int local = (int)blockIdx.x - w.first_block;
    glds_body<false, false, EPI_PARTIAL, 4>(p, w.tiles_n, w.nwg_mn, w.k_chunk, w.part, w.rs_part, w.mode, w.part_bf16,
                                            local % w.grid_x, local / w.grid_x);
}

struct GroupQueue {
    int scope = 0;
    std::vector<WgItem> q;
};
struct GroupQueues {
    std::mutex mu;
    std::map<hipStream_t, GroupQueue> by_stream;
};
GroupQueues& group_queues() { static GroupQueues g; return g; }

int group_launch_locked(GroupQueue& gq, hipStream_t st) {
    size_t at = 0;
    while (at < gq.q.size()) {
        WgTable t;
        int blocks = 0, k = 0;
        for (; k < GROUP_MAX && at < gq.q.size(); ++k, ++at) {
            t.it[k] = gq.q[at];
            t.it[k].first_block = blocks;
            blocks += gq.q[at].first_block;         // (queued with its block count in this field)
        }
        t.n = k;
        const size_t lds = (size_t)4 * 2 * IMG * sizeof(bf16_t);
        static bool once = false;
        if (!once) {
            (void)hipFuncSetAttribute((const void*)gemm_bf16_wgrad_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds);
            once = true;
        }
        hipLaunchKernelGGL(gemm_bf16_wgrad_group_kernel, dim3(blocks), dim3(256), lds, st, t);
    }
    gq.q.clear();
    return 0;
}

template <bool AKC, bool BKC, int EPI>
void launch(const dsvg_gemm_desc& d, dim3 grid, int tiles_n, int nwg, int k_chunk, float* part, float* rs_part, int mode,
            int nst, hipStream_t st, int pbf = 0) {
    static const int occ = getenv("DSVG_GEMM_OCC") ? atoi(getenv("DSVG_GEMM_OCC")) : 4;
    static const int w_warm = getenv("DSVG_GEMM_WARM") ? atoi(getenv("DSVG_GEMM_WARM")) : 0;      // A/B knob (round 6: measured without effect on the step, off): weight operand into L2 up front
    if (EPI != EPI_PARTIAL && w_warm) pbf |= 16;
    if (nst == 4) {
        const size_t lds = (size_t)4 * 2 * IMG * sizeof(bf16_t);
        static bool once = false;
        if (!once) {
            (void)hipFuncSetAttribute((const void*)gemm_bf16_glds_kernel<AKC, BKC, EPI, 4, 4>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            once = true;
        }
        hipLaunchKernelGGL((gemm_bf16_glds_kernel<AKC, BKC, EPI, 4, 4>), grid, dim3(256), lds, st, d, tiles_n, nwg, k_chunk,
                           part, rs_part, mode, pbf);
    } else if (nst == 2)
        hipLaunchKernelGGL((gemm_bf16_glds_kernel<AKC, BKC, EPI, 2, 4>), grid, dim3(256), 0, st, d, tiles_n, nwg, k_chunk,
                           part, rs_part, mode, pbf);
    else if (occ == 5)
        hipLaunchKernelGGL((gemm_bf16_glds_kernel<AKC, BKC, EPI, 1, 5>), grid, dim3(256), 0, st, d, tiles_n, nwg, k_chunk,
                           part, rs_part, mode, pbf);
    else
        hipLaunchKernelGGL((gemm_bf16_glds_kernel<AKC, BKC, EPI, 1, 4>), grid, dim3(256), 0, st, d, tiles_n, nwg, k_chunk,
                           part, rs_part, mode, pbf);
}

}  // namespace

bool dsvg_gemm_bf16_glds_try(const dsvg_gemm_desc& d, int epi, dim3 grid, int tiles_n, int nwg, int k_chunk, float* part,
                             float* rs_part, int mode, hipStream_t st, int* part_is_bf16) {
    static const bool pbf_on = !(getenv("DSVG_SPLITK_BF16") && atoi(getenv("DSVG_SPLITK_BF16")) == 0);      // A/B knob
    static const bool disabled = getenv("DSVG_GEMM_NOGLDS") != nullptr;         // A/B knob
    static const int nst_env = getenv("DSVG_GEMM_STAGES") ? atoi(getenv("DSVG_GEMM_STAGES")) : 1;
    if (disabled || d.impl == 2 || mode == 2 || d.a_drop_p > 0.f) return false;
    // launches of at most one workgroup per CU (the 4096-row group stages: 64-192 workgroups) take the 4-stage variant:
    // nothing else on the CU hides a DMA round trip per K step there.  impl 6 forces it (tests), DSVG_GEMM_DEEP_WGS = 0 disables
    static const int deep_wgs = getenv("DSVG_GEMM_DEEP_WGS") ? atoi(getenv("DSVG_GEMM_DEEP_WGS")) : 256;
    const long long n_wgs = (long long)grid.x * grid.y * grid.z;
    const int nst = d.impl == 6 ? 4 : (d.impl == 3 ? 2 : (d.impl == 4 ? 1 : (n_wgs <= deep_wgs ? 4 : nst_env)));
    if ((d.K % GBK) || (part && (k_chunk % GBK))) return false;
    if ((d.lda & 7) || (d.ldb & 7) || ((uintptr_t)d.A & 15) || ((uintptr_t)d.B & 15)) return false;
    // token-major (mn-contiguous) operands are read in 8-element chunks: the row must be padded to a multiple of 8
    // elements (checked by the caller) - a ragged M is fine, a ragged N only for the k-contiguous B of the forward GEMMs
    if (!d.a_kc && d.M < 8) return false;
    if (!d.b_kc && ((d.N & 7) || d.N < 8)) return false;
    if (d.M < 1 || d.N < 1) return false;
    // per-lane source offsets are 32-bit byte offsets
    const size_t span_a = d.a_kc ? (size_t)d.M * d.lda : (size_t)GBK * d.lda + d.M;
    const size_t span_b = d.b_kc ? (size_t)d.N * d.ldb : (size_t)GBK * d.ldb + d.N;
    if (span_a * 2 >= (1ull << 32) || span_b * 2 >= (1ull << 32)) return false;
    if (part) {
        if (((uintptr_t)part & 15) || (d.N & 3)) return false;
        if (!d.a_kc && !d.b_kc) {
            const int pbf = (pbf_on && part_is_bf16 && !(d.N & 7)) ? 1 : 0;
            if (nst == 4 && grid.z == 1) {      // an open group scope on this stream: queue instead of launching
                GroupQueues& G = group_queues();
                std::lock_guard<std::mutex> lk(G.mu);
                auto it = G.by_stream.find(st);
                if (it != G.by_stream.end() && it->second.scope) {
                    WgItem w;
                    w.A = d.A; w.B = d.B; w.part = part; w.rs_part = rs_part;
                    w.M = d.M; w.N = d.N; w.K = d.K; w.tiles_n = tiles_n; w.nwg_mn = nwg; w.k_chunk = k_chunk; w.mode = mode;
                    w.part_bf16 = pbf; w.first_block = (int)(grid.x * grid.y); w.grid_x = (int)grid.x;
                    w.lda = d.lda; w.ldb = d.ldb;
                    it->second.q.push_back(w);
                    if (part_is_bf16) *part_is_bf16 = pbf;
                    return true;
                }
            }
            launch<false, false, EPI_PARTIAL>(d, grid, tiles_n, nwg, k_chunk, part, rs_part, mode, nst, st, pbf);
            if (part_is_bf16) *part_is_bf16 = pbf;
            return true;
        }
        return false;
    }
    if (d.bias && ((uintptr_t)d.bias & 3)) return false;
    if (d.a_kc && d.b_kc) {
        if (epi == EPI_BIAS) launch<true, true, EPI_BIAS>(d, grid, tiles_n, nwg, k_chunk, part, rs_part, mode, nst, st);
        else if (epi == EPI_BIAS_RES_DROP) launch<true, true, EPI_BIAS_RES_DROP>(d, grid, tiles_n, nwg, k_chunk, part, rs_part, mode, nst, st);
        else if (epi == EPI_BIAS_RELU_DROP) launch<true, true, EPI_BIAS_RELU_DROP>(d, grid, tiles_n, nwg, k_chunk, part, rs_part, mode, nst, st);
        else return false;
        return true;
    }
    if (d.a_kc && !d.b_kc) {
        if (epi == EPI_BIAS) launch<true, false, EPI_BIAS>(d, grid, tiles_n, nwg, k_chunk, part, rs_part, mode, nst, st);
        else if (epi == EPI_GATE) launch<true, false, EPI_GATE>(d, grid, tiles_n, nwg, k_chunk, part, rs_part, mode, nst, st);
        else return false;
        return true;
    }
    return false;
}

// launch what is queued on `st` (no-op without a queue); called when a group scope closes and, from gemm.hip, before any
// deferred reduction of that stream runs (a reduction must never overtake the GEMM that produces its partials)
int dsvg_gemm_group_flush(hipStream_t st) {
    GroupQueues& G = group_queues();
    std::lock_guard<std::mutex> lk(G.mu);
    auto it = G.by_stream.find(st);
    if (it == G.by_stream.end() || it->second.q.empty()) return 0;
    const int rc = group_launch_locked(it->second, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { dsvg_set_error("gemm group: launch failed: %s", hipGetErrorString(e)); return -2; }
    return rc;
}

extern "C" int dsvg_gemm_group_scope(int32_t on, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    {
        GroupQueues& G = group_queues();
        std::lock_guard<std::mutex> lk(G.mu);
        if (on) { G.by_stream[st].scope = 1; return 0; }
        auto it = G.by_stream.find(st);
        if (it == G.by_stream.end()) return 0;
        it->second.scope = 0;
    }
    const int rc = dsvg_gemm_group_flush(st);
    {
        GroupQueues& G = group_queues();
        std::lock_guard<std::mutex> lk(G.mu);
        auto it = G.by_stream.find(st);
        if (it != G.by_stream.end() && it->second.q.empty() && !it->second.scope) G.by_stream.erase(it);
    }
    return rc;
}
