// Shared GEMM epilogue:  C = [C +] [res +] drop( gate( act( acc + bias [+ res if res_pre] ) ) )
#pragma once
#include "dsvg_common.h"
#include "../../include/dsvg.h"

template <typename T>
__device__ __forceinline__ void gemm_epilogue(const dsvg_gemm_desc& p, const DropCtx& dc, int m, int n, float acc) {
    float v = acc;
    if (p.bias) v += p.bias[n];
    if (p.res && p.res_pre) v += Elem<T>::ld((const T*)p.res + (size_t)m * p.ldres + n);
    if (p.act == 1) v = fmaxf(v, 0.f);
    if (p.gate) {
        const float g = Elem<T>::ld((const T*)p.gate + (size_t)m * p.ldgate + n);
        v = g > 0.f ? v * p.gate_scale : 0.f;
    }
    v *= drop_mult(dc, (uint64_t)m * p.N + n);
    if (p.res && !p.res_pre) v += Elem<T>::ld((const T*)p.res + (size_t)m * p.ldres + n);
    if (p.c_f32) {
        float* c = (float*)p.C + (size_t)m * p.ldc + n;
        if (p.accumulate) v += *c;
        *c = v;
    } else {
        T* c = (T*)p.C + (size_t)m * p.ldc + n;
        if (p.accumulate) v += Elem<T>::ld(c);
        Elem<T>::st(c, v);
    }
}
