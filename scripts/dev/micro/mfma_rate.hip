// Dev microbenchmark: what one wave per SIMD sustains on v_mfma_f32_32x32x16_f16 with random operands on every CU
// (cycles per MFMA by s_memtime, in-kernel clock by s_memrealtime).  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const f16x8* in, float* out, long long* stamps, int iters, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    f16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(blockIdx.x * 8 + i) * 64 + lane]; b[i] = in[(4096 + i) * 64 + lane]; }
    f16x8* l = reinterpret_cast<f16x8*>(smem) + (threadIdx.x >> 6) * 512;
    for (int i = 0; i < 8; ++i) l[i * 64 + lane] = a[i];
    __syncthreads();
    f32x16 acc0 = {0}, acc1 = {0};
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (mode == 1) { for (int i = 0; i < 8; ++i) a[i] = l[i * 64 + ((lane + it) & 63)]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[i], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i + 1) & 7], acc1, 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}
int main(int argc, char** argv) {
    const int iters = 20000;
    std::vector<_Float16> h((4096 + 8) * 64 * 8);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.25f);
    f16x8* in; float* out; long long* st;
    hipMalloc(&in, h.size() * 2); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&st, 256 * 16);
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int waves : {4, 8}) for (int mode : {0, 1}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (waves == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 65536, 0, in, out, st, iters, mode);
            else hipLaunchKernelGGL(k<8>, dim3(256), dim3(512), 65536, 0, in, out, st, iters, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long hs[512]; hipMemcpy(hs, st, 256 * 16, hipMemcpyDeviceToHost);
            double cyc = 0, real = 0; for (int i = 0; i < 256; ++i) { cyc += hs[2 * i]; real += hs[2 * i + 1]; }
            cyc /= 256; real /= 256;
            const double mfmas = (double)iters * 16;
            const double flop = 256.0 * waves * mfmas * 32768.0;
            printf("waves/WG %d mode %d (0 regs, 1 LDS reads): %.3f ms  %.1f TFLOP/s  cycles/MFMA per wave %.1f  clock %.2f GHz\n", waves, mode, ms,
                   flop / (ms * 1e-3) / 1e12, cyc / mfmas, cyc / real * 0.1);
        }
    }
    return 0;
}
