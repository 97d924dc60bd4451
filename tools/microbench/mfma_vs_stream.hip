// Do v_mfma_f32_16x16x4_f32 and a device-memory stream slow each other down when they share the CUs (different waves, no data
// dependence between them)?  Workgroups of 8 waves: waves 0-3 (one per SIMD) run MFMA chains with changing operands for a fixed
// number of iterations, waves 4-7 copy a large buffer (16-byte lanes) for as long as the MFMA waves run ... or do nothing.
// Prints the MFMA rate alone, the copy rate alone, and both rates when they run together.
#include <hip/hip_runtime.h>
#include <cstdio>
using f4 = __attribute__((ext_vector_type(4))) float;
constexpr int ACC = 9;
__global__ __launch_bounds__(512) void mix(float* out, const f4* src, f4* dst, size_t n4, int iters, int do_mfma, int do_copy, unsigned long long* moved) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        if (!do_mfma) return;
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
    } else {
        if (!do_copy) return;
        // a fixed amount per wave when alone; the same amount when mixed (the host sizes iters so that both take similar time alone)
        const size_t waves = (size_t)gridDim.x * 4, w = (size_t)blockIdx.x * 4 + (wave - 4);
        const size_t per = n4 / waves / 64 * 64;
        const f4* s = src + w * per;
        f4* d = dst + w * per;
        for (size_t i = lane; i < per; i += 256) {          // four 1-KB loads in flight per wave
            f4 v0 = s[i], v1 = i + 64 < per ? s[i + 64] : v0, v2 = i + 128 < per ? s[i + 128] : v0, v3 = i + 192 < per ? s[i + 192] : v0;
            d[i] = v0;
            if (i + 64 < per) d[i + 64] = v1;
            if (i + 128 < per) d[i + 128] = v2;
            if (i + 192 < per) d[i + 192] = v3;
        }
        if (lane == 0) atomicAdd(moved, (unsigned long long)per * 32);
    }
}
int main() {
    const size_t bytes = (size_t)1 << 30, n4 = bytes / 16;
    f4 *src, *dst; float* out; unsigned long long* moved;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess || hipMalloc(&moved, 8) != hipSuccess) return 1;
    (void)hipMemset(src, 1, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 2;      // two workgroups per CU: 2 MFMA waves + 2 copy waves per SIMD
    auto run = [&](int iters, int m, int c, const char* tag) {
        (void)hipMemset(moved, 0, 8);
        hipLaunchKernelGGL(mix, dim3(blocks), dim3(512), 0, 0, out, src, dst, n4, iters, m, c, moved);
        (void)hipMemset(moved, 0, 8);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(mix, dim3(blocks), dim3(512), 0, 0, out, src, dst, n4, iters, m, c, moved);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long mv = 0; (void)hipMemcpy(&mv, moved, 8, hipMemcpyDeviceToHost);
        const double tf = m ? (double)blocks * 4 * iters * 4.0 * ACC * 2048 / (ms * 1e-3) / 1e12 : 0.0;
        printf("%-28s %.3f ms   MFMA %.1f TFLOP/s   copy %.2f TB/s (read + write)\n", tag, ms, tf, mv / (ms * 1e-3) / 1e12);
        return ms;
    };
    const float t_copy = run(0, 0, 1, "copy alone (1 GiB -> 1 GiB)");
    int iters = 2000;
    float t_m = run(iters, 1, 0, "MFMA alone");
    iters = (int)(iters * t_copy / t_m);               // same duration as the copy
    run(iters, 1, 0, "MFMA alone (matched time)");
    run(iters, 1, 1, "both together");
    return 0;
}
