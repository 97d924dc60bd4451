// Sustained rate of v_mfma_f32_16x16x4_f32 (the exact-fp32 MFMA every Linear kernel of this library uses) with nothing else going on:
// W waves per SIMD, each issuing ACC independent accumulator chains.  Prints TFLOP/s and the implied cycles per MFMA at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f4 = __attribute__((ext_vector_type(4))) float;
// the same with operands that change every iteration and differ per lane (random mantissas, exponents near 1): data-dependent power
template <int ACC>
__global__ __launch_bounds__(1024) void spin_data(float* out, int iters) {
    f4 acc[ACC];
#pragma unroll
    for (int q = 0; q < ACC; ++q) acc[q] = f4{0.f, 0.f, 0.f, 0.f};
    unsigned h = (threadIdx.x + 1) * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            h = h * 1664525u + 1013904223u;
            const float a = __uint_as_float(0x3f800000u | (h >> 9)) - 1.5f, b = __uint_as_float(0x3f800000u | ((h * 2246822519u) >> 9)) - 1.5f;
#pragma unroll
            for (int q = 0; q < ACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
        }
    }
    float r = 0.f;
#pragma unroll
    for (int q = 0; q < ACC; ++q) r += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    if (r == 12345.678f) out[0] = r;
}
template <int ACC>
__global__ __launch_bounds__(1024) void spin(float* out, int iters) {
    f4 acc[ACC];
#pragma unroll
    for (int q = 0; q < ACC; ++q) acc[q] = f4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int q = 0; q < ACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
    }
    float r = 0.f;
#pragma unroll
    for (int q = 0; q < ACC; ++q) r += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    if (r == 12345.678f) out[0] = r;
}
// sixteen accumulators, 64 + 4 DISTINCT operand registers (the register pattern of the GEMM kernels' inner loop, nothing else in the loop);
// LDS != 0: the 16 A-operand quads are re-read from LDS every iteration with ds_read_b128, as the GEMM kernels do
template <int LDS>
__global__ __launch_bounds__(512) void spin_regs(float* out, int iters) {
    __shared__ float tile[16 * 16 * 20];
    f4 acc[16], wv[16];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 16 * 20; i += blockDim.x) tile[i] = 1e-3f * (i % 97);
    __syncthreads();
    const float* wl = tile + (lane & 15) * 20 + 4 * (lane >> 4);
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc[q] = f4{0.f, 0.f, 0.f, 0.f}; wv[q] = *reinterpret_cast<const f4*>(wl + 16 * q * 20); }
    f4 xv = f4{1.f + lane, 2.f, 3.f, 4.f};
    for (int i = 0; i < iters; ++i) {
        if (LDS) {
#pragma unroll
            for (int q = 0; q < LDS; ++q) wv[q] = *reinterpret_cast<const volatile f4*>(wl + 16 * q * 20);
        }
#pragma unroll
        for (int q0 = 0; q0 < 16; q0 += 4)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[q0 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[q0 + j][s], xv[s], acc[q0 + j], 0, 0, 0);
        xv[0] += 1e-6f;
    }
    float r = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) r += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    if (r == 12345.678f) out[0] = r;
}
template <int LDS>
void run_regs(int waves_per_simd, int iters) {
    float* out; (void)hipMalloc(&out, 4);
    const int threads = 64 * 4 * waves_per_simd > 512 ? 512 : 64 * 4 * waves_per_simd;
    const int blocks = 256 * (64 * 4 * waves_per_simd / threads);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(spin_regs<LDS>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(spin_regs<LDS>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double mfmas = (double)blocks * (threads / 64) * iters * 64.0;
    printf("distinct operand registers%s, waves/SIMD=%d: %.3f ms  %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n",
           LDS == 16 ? " + 16 ds_read_b128 per 64 MFMAs" : (LDS == 8 ? " + 8 ds_read_b128 per 64 MFMAs" : (LDS == 4 ? " + 4 ds_read_b128 per 64 MFMAs" : "")), waves_per_simd, ms, mfmas * 2048 / (ms * 1e-3) / 1e12, 1024.0 * 2.4e9 * ms * 1e-3 / mfmas);
    (void)hipFree(out);
}
template <int ACC>
void run(int waves_per_simd, int iters, bool data = false) {
    float* out; hipMalloc(&out, 4);
    const int threads = 64 * 4 * waves_per_simd > 1024 ? 1024 : 64 * 4 * waves_per_simd;
    const int blocks = 256 * (64 * 4 * waves_per_simd / threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
        if (data) hipLaunchKernelGGL(spin_data<ACC>, dim3(blocks), dim3(threads), 0, 0, out, iters);
        else hipLaunchKernelGGL(spin<ACC>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    };
    launch();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double mfmas = (double)blocks * (threads / 64) * iters * 4.0 * ACC;
    const double tf = mfmas * 2048 / (ms * 1e-3) / 1e12;
    printf("%s ACC=%d waves/SIMD=%d: %.3f ms  %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", data ? "random operands  " : "constant operands", ACC, waves_per_simd, ms, tf,
           1024.0 * 2.4e9 * ms * 1e-3 / mfmas);
    hipFree(out);
}
// ---- v_mfma_f32_32x32x2_f32: the same peak (64 cycles per instruction), twice the MACs per operand register -----------------------------
using f16v = __attribute__((ext_vector_type(16))) float;
template <int ACC>
__global__ __launch_bounds__(512) void spin32(float* out, int iters) {
    f16v acc[ACC];
#pragma unroll
    for (int q = 0; q < ACC; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int q = 0; q < ACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
    }
    float r = 0.f;
#pragma unroll
    for (int q = 0; q < ACC; ++q) r += acc[q][0] + acc[q][7] + acc[q][15];
    if (r == 12345.678f) out[0] = r;
}
// the register pattern of a 64-row x 128-column tile per wave on 32x32x2: 2 x 4 accumulator tiles, per 8-k block (four instructions per
// tile) 2 + 4 operand quads, re-read from LDS with ds_read_b128 when LDS != 0 (6 reads per 32 MFMAs = the flops of 64 16x16x4 ones)
template <int LDS>
__global__ __launch_bounds__(256) void spin_regs32(float* out, int iters) {
    __shared__ float tile[6 * 32 * 12];
    f16v acc[2][4];
    f4 av[2], bv[4];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 6 * 32 * 12; i += blockDim.x) tile[i] = 1e-3f * (i % 97);
    __syncthreads();
    const float* tl = tile + (lane & 31) * 12 + 4 * (lane >> 5);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        av[a] = *reinterpret_cast<const f4*>(tl + a * 32 * 12);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) bv[b] = *reinterpret_cast<const f4*>(tl + (2 + b) * 32 * 12);
    for (int i = 0; i < iters; ++i) {
        if (LDS) {
#pragma unroll
            for (int a = 0; a < 2; ++a) av[a] = *reinterpret_cast<const volatile f4*>(tl + a * 32 * 12);
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[b] = *reinterpret_cast<const volatile f4*>(tl + (2 + b) * 32 * 12);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][s], bv[b][s], acc[a][b], 0, 0, 0);
        av[0][0] += 1e-6f;
    }
    float r = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) r += acc[a][b][0] + acc[a][b][9];
    if (r == 12345.678f) out[0] = r;
}
template <int ACC>
void run32(int waves_per_simd, int iters) {
    float* out; (void)hipMalloc(&out, 4);
    const int threads = 64 * 4 * waves_per_simd > 512 ? 512 : 64 * 4 * waves_per_simd;
    const int blocks = 256 * (64 * 4 * waves_per_simd / threads);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(spin32<ACC>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(spin32<ACC>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double mfmas = (double)blocks * (threads / 64) * iters * 4.0 * ACC;
    printf("32x32x2 constant operands ACC=%d waves/SIMD=%d: %.3f ms  %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", ACC, waves_per_simd, ms,
           mfmas * 4096 / (ms * 1e-3) / 1e12, 1024.0 * 2.4e9 * ms * 1e-3 / mfmas);
    (void)hipFree(out);
}
template <int LDS>
void run_regs32(int waves_per_simd, int iters) {
    float* out; (void)hipMalloc(&out, 4);
    const int threads = 64 * 4 * waves_per_simd > 256 ? 256 : 64 * 4 * waves_per_simd;
    const int blocks = 256 * (64 * 4 * waves_per_simd / threads);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(spin_regs32<LDS>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(spin_regs32<LDS>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double mfmas = (double)blocks * (threads / 64) * iters * 32.0;
    printf("32x32x2, 2 x 4 tiles per wave%s, waves/SIMD=%d: %.3f ms  %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n",
           LDS ? " + 6 ds_read_b128 per 32 MFMAs" : ", operands in registers", waves_per_simd, ms, mfmas * 4096 / (ms * 1e-3) / 1e12,
           1024.0 * 2.4e9 * ms * 1e-3 / mfmas);
    (void)hipFree(out);
}

int main() {
    for (int w : {1, 2}) { run32<2>(w, 8000); run32<8>(w, 4000); }
    for (int w : {1, 2}) { run_regs32<0>(w, 4000); run_regs32<1>(w, 4000); }

    for (int w : {1, 2, 4}) { run<3>(w, 20000); run<9>(w, 8000); }
    run<9>(4, 200000);      // ~0.3 s: sustained
    for (int w : {1, 2}) { run_regs<0>(w, 4000); run_regs<16>(w, 4000); run_regs<8>(w, 4000); run_regs<4>(w, 4000); }
    run<9>(2, 8000, true);
    run<9>(4, 8000, true);
    run<9>(4, 200000, true);
    return 0;
}
