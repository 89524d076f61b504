// HBM-bound row kernels: L2-normalise, row-norm max, embedding-row gather (gfx950).
//
//   k_normalize_rows  <- `normalize` (reference util_amazon_filtered.py:28-31; the
//                        ||v||+1e-4 variant of fine_tune_ours.py:38-40 is rule 1)
//   k_gather_rows     <- NodeAsinEmbedding.forward (reference model/NodeEmbedding.py:137-138)
// Each row is owned by a group of LPR lanes that move 16 bytes per lane per access (coalesced
// 16 B x LPR segments); reductions are butterflies inside the lane group.
#include "sss_common.h"

namespace sss {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int LPR>
__global__ __launch_bounds__(256) void k_normalize_rows(float* __restrict__ x, long n, int d, long ld,
                                                        float eps, int rule) {
    const int sub = threadIdx.x % LPR;
    const long rows_per_block = 256 / LPR;
    const int nv = d / 4;
    for (long row = (long)blockIdx.x * rows_per_block + threadIdx.x / LPR; row < n;
         row += (long)gridDim.x * rows_per_block) {
        float4* p = reinterpret_cast<float4*>(x + row * ld);
        float ss = 0.f;
        for (int i = sub; i < nv; i += LPR) {
            const float4 v = p[i];
            ss += v.x * v.x; ss += v.y * v.y; ss += v.z * v.z; ss += v.w * v.w;
        }
        ss = group_sum<LPR>(ss);
        const float den = rule == 0 ? sqrtf(fmaxf(ss, eps)) : sqrtf(ss) + eps;
        for (int i = sub; i < nv; i += LPR) {
            float4 v = p[i];
            v.x /= den; v.y /= den; v.z /= den; v.w /= den;
            p[i] = v;
        }
    }
}

template <int LPR>
__global__ __launch_bounds__(256) void k_row_norm_max(const float* __restrict__ x, long n, int d,
                                                      float* __restrict__ out) {
    const int sub = threadIdx.x % LPR;
    const long rows_per_block = 256 / LPR;
    const int nv = d / 4;
    float m = 0.f;
    for (long row = (long)blockIdx.x * rows_per_block + threadIdx.x / LPR; row < n;
         row += (long)gridDim.x * rows_per_block) {
        const float4* p = reinterpret_cast<const float4*>(x + row * (long)d);
        float ss = 0.f;
        for (int i = sub; i < nv; i += LPR) {
            const float4 v = p[i];
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        ss = group_sum<LPR>(ss);
        m = fmaxf(m, ss);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    // non-negative floats order like their bit patterns
    if ((threadIdx.x & 63) == 0)
        atomicMax(reinterpret_cast<unsigned int*>(out), __builtin_bit_cast(unsigned int, sqrtf(m) * 1.0000002f));
}

// bf16 rows: max over rows of the 2-norm of the (exactly converted) float32 values
__global__ __launch_bounds__(256) void k_row_norm_max_bf16(const unsigned short* __restrict__ x, long n, int d,
                                                           float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    float m = 0.f;
    for (long row = wave; row < n; row += nwaves) {
        const u32x4* p = reinterpret_cast<const u32x4*>(x + row * (long)d);
        float ss = 0.f;
        for (int i = lane; i < d / 8; i += 64) {
            const u32x4 v = p[i];
#define SSS_SQ2(w)                                                                   \
    { const float lo = __builtin_bit_cast(float, v.w << 16), hi = __builtin_bit_cast(float, v.w & 0xFFFF0000u); \
      ss += lo * lo + hi * hi; }
            SSS_SQ2(x) SSS_SQ2(y) SSS_SQ2(z) SSS_SQ2(w)
#undef SSS_SQ2
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        m = fmaxf(m, ss);
    }
    if (lane == 0)
        atomicMax(reinterpret_cast<unsigned int*>(out), __builtin_bit_cast(unsigned int, sqrtf(m) * 1.0000002f));
}

// float32 -> bfloat16, round to nearest even (plain cast: v_cvt_pk_bf16_f32, NaN stays NaN);
// 8 elements per thread, 32-byte loads / 16-byte stores.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void k_f32_to_bf16(const float* __restrict__ x, long n8, unsigned short* __restrict__ y) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const f32x4 a = reinterpret_cast<const f32x4*>(x)[2 * i], b = reinterpret_cast<const f32x4*>(x)[2 * i + 1];
        bf16x8_t o;
        o[0] = (__bf16)a.x; o[1] = (__bf16)a.y; o[2] = (__bf16)a.z; o[3] = (__bf16)a.w;
        o[4] = (__bf16)b.x; o[5] = (__bf16)b.y; o[6] = (__bf16)b.z; o[7] = (__bf16)b.w;
        reinterpret_cast<bf16x8_t*>(y)[i] = o;
    }
}

// f32 row [d] -> [hi(d) | lo(d)] bfloat16 (scan.h: DT_SPLIT): hi = rne(x), lo = rne(x - hi); a lo that
// is not finite (x = +-inf, or hi rounded up to inf) is stored as 0.  8 elements per thread.
__global__ __launch_bounds__(256) void k_split_bf16(const float* __restrict__ x, long n, int d8,
                                                    unsigned short* __restrict__ y) {
    const long total = n * d8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / d8;
        const int j = (int)(i - row * d8);
        const f32x4 a = reinterpret_cast<const f32x4*>(x)[2 * i], b = reinterpret_cast<const f32x4*>(x)[2 * i + 1];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        bf16x8_t hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            hi[e] = (__bf16)v[e];
            const float rem = v[e] - (float)hi[e];
            lo[e] = (__bf16)(__builtin_isfinite(rem) ? rem : 0.f);
        }
        bf16x8_t* out = reinterpret_cast<bf16x8_t*>(y) + row * 2 * d8;
        out[j] = hi;
        out[d8 + j] = lo;
    }
}

// max |x_i| over `count` contiguous floats -> *out (atomic max on the bit pattern of a non-negative
// float; caller zeroes).  NaNs are ignored (fmaxf), +-inf gives inf.  count % 4 == 0.
__global__ __launch_bounds__(256) void k_abs_max(const float* __restrict__ x, long n4, float* __restrict__ out) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 a = reinterpret_cast<const f32x4*>(x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(a.x), fabsf(a.y))), fmaxf(fabsf(a.z), fabsf(a.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(m));
}

// f32 -> float16 of x * 2^shift (scan.h: DT_F16), round to nearest even; 8 elements per thread.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void k_scale_f16(const float* __restrict__ x, long n8, int shift,
                                                   unsigned short* __restrict__ y) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const f32x4 a = reinterpret_cast<const f32x4*>(x)[2 * i], b = reinterpret_cast<const f32x4*>(x)[2 * i + 1];
        f16x8_t o;
        o[0] = (_Float16)ldexpf(a.x, shift); o[1] = (_Float16)ldexpf(a.y, shift);
        o[2] = (_Float16)ldexpf(a.z, shift); o[3] = (_Float16)ldexpf(a.w, shift);
        o[4] = (_Float16)ldexpf(b.x, shift); o[5] = (_Float16)ldexpf(b.y, shift);
        o[6] = (_Float16)ldexpf(b.z, shift); o[7] = (_Float16)ldexpf(b.w, shift);
        reinterpret_cast<f16x8_t*>(y)[i] = o;
    }
}

// max over rows of || y_i * 2^-shift - x_i ||_2 (y the scaled f16 image of x) -> *out, atomically
// maximised (caller zeroes): the corpus residual of the DT_F16 error bound.  One wave per row.
__global__ __launch_bounds__(256) void k_f16_resid_max(const float* __restrict__ x, const _Float16* __restrict__ y,
                                                       long n, int d, int shift, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    float m = 0.f;
    for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += (long)gridDim.x * 4) {
        double s = 0.0;
        for (int kk = lane; kk < d; kk += 64) {
            const double r = (double)ldexpf((float)y[row * d + kk], -shift) - (double)x[row * d + kk];
            s += r * r;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float f = (float)(sqrt(s) * (1.0 + 1e-6));       // rounded up
        if (f == f) m = fmaxf(m, f);
    }
    if (lane == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(m));
}

template <int LPR>
__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ table,
                                                     const long* __restrict__ ids, long n, int d,
                                                     float* __restrict__ out, long ld_out) {
    const int sub = threadIdx.x % LPR;
    const long rows_per_block = 256 / LPR;
    const int nv = d / 4;
    for (long row = (long)blockIdx.x * rows_per_block + threadIdx.x / LPR; row < n;
         row += (long)gridDim.x * rows_per_block) {
        const float4* src = reinterpret_cast<const float4*>(table + ids[row] * (long)d);
        float4* dst = reinterpret_cast<float4*>(out + row * ld_out);
        for (int i = sub; i < nv; i += LPR) dst[i] = src[i];
    }
}

// use_id_embedding=True of the reference encoder (model/model.py:288-289): embedding['product'] =
// concat(id_embedding(x), text_features) -- out[i] = [table[ids[i]] (d_id floats) | feat[i] (d_f floats) | pad zeros],
// one pass, one lane group per row.  feat == nullptr writes zeros there (query rows padded to the product width).
template <int LPR>
__global__ __launch_bounds__(256) void k_gather_concat_rows(const float* __restrict__ table, const long* __restrict__ ids,
                                                            int d_id, const float* __restrict__ feat, long ld_feat, int d_f,
                                                            int d_pad, long n, float* __restrict__ out, long ld_out) {
    const int sub = threadIdx.x % LPR;
    const long rows_per_block = 256 / LPR;
    const int nv_id = d_id / 4, nv_f = d_f / 4, nv_pad = d_pad / 4;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long row = (long)blockIdx.x * rows_per_block + threadIdx.x / LPR; row < n;
         row += (long)gridDim.x * rows_per_block) {
        float4* dst = reinterpret_cast<float4*>(out + row * ld_out);
        if (nv_id) {
            const float4* src = reinterpret_cast<const float4*>(table + ids[row] * (long)d_id);
            for (int i = sub; i < nv_id; i += LPR) dst[i] = src[i];
        }
        if (feat) {
            const float4* f = reinterpret_cast<const float4*>(feat + row * ld_feat);
            for (int i = sub; i < nv_f; i += LPR) dst[nv_id + i] = f[i];
        } else {
            for (int i = sub; i < nv_f; i += LPR) dst[nv_id + i] = zero;
        }
        for (int i = sub; i < nv_pad; i += LPR) dst[nv_id + nv_f + i] = zero;
    }
}

static int lanes_per_row(int d) {
    const int nv = d / 4;
    int l = 1;
    while (l < nv && l < 64) l <<= 1;
    return l;
}
static unsigned grid_for(long n, int lpr) {
    const long rpb = 256 / lpr;
    long g = (n + rpb - 1) / rpb;
    if (g > 256 * 8) g = 256 * 8;   // ~8 blocks per CU, grid-stride the rest
    return (unsigned)(g < 1 ? 1 : g);
}

#define SSS_DISPATCH_LPR(lpr, CALL) \
    switch (lpr) {                  \
        case 1: { constexpr int L = 1; CALL; } break;   \
        case 2: { constexpr int L = 2; CALL; } break;   \
        case 4: { constexpr int L = 4; CALL; } break;   \
        case 8: { constexpr int L = 8; CALL; } break;   \
        case 16: { constexpr int L = 16; CALL; } break; \
        case 32: { constexpr int L = 32; CALL; } break; \
        default: { constexpr int L = 64; CALL; } break; \
    }

int normalize_rows(float* x, long n, int d, long ld, float eps, int rule, hipStream_t st) {
    if (n < 0 || d <= 0 || d % 4 || ld < d || ld % 4 || (rule != 0 && rule != 1)) {
        set_error("normalize_rows: need n >= 0, d %% 4 == 0, ld >= d, ld %% 4 == 0, rule in {0,1}");
        return SSS_EINVAL;
    }
    if (n == 0) return SSS_OK;
    const int lpr = lanes_per_row(d);
    SSS_DISPATCH_LPR(lpr, hipLaunchKernelGGL(k_normalize_rows<L>, dim3(grid_for(n, L)), dim3(256), 0, st, x, n, d, ld, eps, rule));
    return check_launch("k_normalize_rows");
}

int row_norm_max(const void* xv, long n, int d, int dtype, float* out, hipStream_t st) {
    if (dtype == 1) {
        if (n < 0 || d <= 0 || d % 8) { set_error("row_norm_max: bf16 needs d %% 8 == 0"); return SSS_EINVAL; }
        if (n == 0) return SSS_OK;
        long blocks = (n + 3) / 4;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(k_row_norm_max_bf16, dim3((unsigned)blocks), dim3(256), 0, st,
                           reinterpret_cast<const unsigned short*>(xv), n, d, out);
        return check_launch("k_row_norm_max_bf16");
    }
    const float* x = reinterpret_cast<const float*>(xv);
    if (dtype != 0 || n < 0 || d <= 0 || d % 4) { set_error("row_norm_max: need dtype in {0,1}, n >= 0, d %% 4 == 0"); return SSS_EINVAL; }
    if (n == 0) return SSS_OK;
    const int lpr = lanes_per_row(d);
    SSS_DISPATCH_LPR(lpr, hipLaunchKernelGGL(k_row_norm_max<L>, dim3(grid_for(n, L)), dim3(256), 0, st, x, n, d, out));
    return check_launch("k_row_norm_max");
}

int f32_to_bf16(const float* x, long count, unsigned short* y, hipStream_t st) {
    if (count < 0 || count % 8) { set_error("f32_to_bf16: element count must be a multiple of 8"); return SSS_EINVAL; }
    if (count == 0) return SSS_OK;
    long blocks = (count / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_f32_to_bf16, dim3((unsigned)blocks), dim3(256), 0, st, x, count / 8, y);
    return check_launch("k_f32_to_bf16");
}

int split_bf16(const float* x, long n, int d, unsigned short* y, hipStream_t st) {
    if (n < 0 || d <= 0 || d % 8) { set_error("split_bf16: need n >= 0, d %% 8 == 0"); return SSS_EINVAL; }
    if (n == 0) return SSS_OK;
    long blocks = (n * (d / 8) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_split_bf16, dim3((unsigned)blocks), dim3(256), 0, st, x, n, d / 8, y);
    return check_launch("k_split_bf16");
}

int abs_max(const float* x, long count, float* out, hipStream_t st) {
    if (count < 0 || count % 4) { set_error("abs_max: element count must be a multiple of 4"); return SSS_EINVAL; }
    if (count == 0) return SSS_OK;
    long blocks = (count / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_abs_max, dim3((unsigned)blocks), dim3(256), 0, st, x, count / 4, out);
    return check_launch("k_abs_max");
}

int scale_f16(const float* x, long count, int shift, unsigned short* y, hipStream_t st) {
    if (count < 0 || count % 8) { set_error("scale_f16: element count must be a multiple of 8"); return SSS_EINVAL; }
    if (shift < -160 || shift > 160) { set_error("scale_f16: shift out of range"); return SSS_EINVAL; }
    if (count == 0) return SSS_OK;
    long blocks = (count / 8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_scale_f16, dim3((unsigned)blocks), dim3(256), 0, st, x, count / 8, shift, y);
    return check_launch("k_scale_f16");
}

int f16_resid_max(const float* x, const unsigned short* y, long n, int d, int shift, float* out, hipStream_t st) {
    if (n < 0 || d <= 0 || shift < -160 || shift > 160) { set_error("f16_resid_max: bad arguments"); return SSS_EINVAL; }
    if (n == 0) return SSS_OK;
    long blocks = (n + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_f16_resid_max, dim3((unsigned)blocks), dim3(256), 0, st, x, reinterpret_cast<const _Float16*>(y), n,
                       d, shift, out);
    return check_launch("k_f16_resid_max");
}

int gather_rows(const float* table, const long* ids, long n, int d, float* out, long ld_out, hipStream_t st) {
    if (n < 0 || d <= 0 || d % 4 || ld_out < d || ld_out % 4) {
        set_error("gather_rows: need n >= 0, d %% 4 == 0, ld_out >= d, ld_out %% 4 == 0");
        return SSS_EINVAL;
    }
    if (n == 0) return SSS_OK;
    const int lpr = lanes_per_row(d);
    SSS_DISPATCH_LPR(lpr, hipLaunchKernelGGL(k_gather_rows<L>, dim3(grid_for(n, L)), dim3(256), 0, st, table, ids, n, d, out, ld_out));
    return check_launch("k_gather_rows");
}

int gather_concat_rows(const float* table, const long* ids, int d_id, const float* feat, long ld_feat, int d_f, int d_pad,
                       long n, float* out, long ld_out, hipStream_t st) {
    if (n < 0 || d_id < 0 || d_f < 0 || d_pad < 0 || d_id % 4 || d_f % 4 || d_pad % 4 || d_id + d_f + d_pad <= 0 ||
        ld_out % 4 || ld_out < d_id + d_f + d_pad || (feat && (ld_feat % 4 || ld_feat < d_f)) || (d_id > 0 && (!table || !ids))) {
        set_error("gather_concat_rows: need widths %% 4 == 0, ld_out >= d_id + d_f + d_pad, 16-byte aligned row strides");
        return SSS_EINVAL;
    }
    if (n == 0) return SSS_OK;
    const int lpr = lanes_per_row(d_id + d_f + d_pad);
    SSS_DISPATCH_LPR(lpr, hipLaunchKernelGGL(k_gather_concat_rows<L>, dim3(grid_for(n, L)), dim3(256), 0, st, table, ids, d_id, feat,
                                             ld_feat, d_f, d_pad, n, out, ld_out));
    return check_launch("k_gather_concat_rows");
}

}  // namespace sss
