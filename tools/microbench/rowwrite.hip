// What does it cost just to WRITE the forward sweep's output at the ZINC-12k shape (275 167 rows x 6 blocks x 70
// floats = 462 MB)?  Variants of the store pattern, no loads, no math.
// Build: hipcc --offload-arch=gfx950 -O3 rowwrite.hip -o rowwrite
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// one wave per row, A stores of F/2 lanes x float2 (node-major [N][A][F])
__global__ void wave_per_row(float* __restrict__ out, int n_rows, int A, int F, float v) {
    int row = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    int lane = threadIdx.x & 63;
    if (2 * lane >= F) return;
    float* p = out + (size_t)row * A * F + 2 * lane;
    for (int a = 0; a < A; ++a) *reinterpret_cast<float2*>(p + a * F) = make_float2(v, v + a);
}
// R consecutive rows per wave
__global__ void wave_per_rows(float* __restrict__ out, int n_rows, int A, int F, int R, float v) {
    int g = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (2 * lane >= F) return;
    for (int r = 0; r < R; ++r) {
        int row = g * R + r;
        if (row >= n_rows) return;
        float* p = out + (size_t)row * A * F + 2 * lane;
        for (int a = 0; a < A; ++a) *reinterpret_cast<float2*>(p + a * F) = make_float2(v, v + a);
    }
}
// tower-major [T][N][A][F/T]: lane's tower block is far away
__global__ void wave_per_row_towers(float* __restrict__ out, int n_rows, int A, int F, int T, float v) {
    int row = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    int lane = threadIdx.x & 63;
    int f = 2 * lane;
    if (f >= F) return;
    int Ft = F / T, t = f / Ft, ft = f - t * Ft;
    float* p = out + ((size_t)t * n_rows + row) * A * Ft + ft;
    for (int a = 0; a < A; ++a) *reinterpret_cast<float2*>(p + a * Ft) = make_float2(v, v + a);
}
// the same bytes as a flat stream
__global__ void flat(float2* __restrict__ out, size_t n, float v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = make_float2(v, v);
}
// workgroup per 16 rows: the 16 x A x F tile is contiguous, written as one stream by 256 threads
__global__ void block_tile(float* __restrict__ out, int n_rows, int A, int F, float v) {
    size_t r0 = (size_t)blockIdx.x * 16;
    size_t rows = n_rows - r0 < 16 ? n_rows - r0 : 16;
    float2* p = reinterpret_cast<float2*>(out + r0 * A * F);
    size_t n = rows * A * F / 2;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) p[i] = make_float2(v, v);
}

// the sweep's staged store: one wave per 4 rows, every row = T tower pieces of K floats (row stride LD >= K inside a tower
// plane), written as 16-byte lanes over all pieces of the row.  K = 84, LD = 84: today's layout (336-byte pieces at 16-byte
// alignment); K = LD = 96: padded pieces, three full 128-byte lines each
__global__ void staged_rows(float* __restrict__ out, int n_rows, int T, int K, int LD, float v) {
    int g = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    const int K4 = K >> 2, total4 = T * K4;
    for (int r = 0; r < 4; ++r) {
        int row = g * 4 + r;
        if (row >= n_rows) return;
        float* orow = out + (size_t)row * LD;
        for (int i4 = lane; i4 < total4; i4 += 64) {
            int q = i4 / K4, c4 = i4 - q * K4;
            *reinterpret_cast<float4*>(orow + (size_t)q * n_rows * LD + 4 * c4) = make_float4(v, v + r, v, v);
        }
    }
}

// the same bytes with the R rows of a wave written TOGETHER per tower: R * K contiguous floats (1344 bytes for R = 4) as
// consecutive 16-byte lanes, i.e. full 128-byte lines apart from the two ends of the run
__global__ void staged_group(float* __restrict__ out, int n_rows, int T, int K, int R, float v) {
    int g = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    int row0 = g * R;
    if (row0 >= n_rows) return;
    int rows = min(R, n_rows - row0);
    const int run4 = rows * K >> 2;
    for (int q = 0; q < T; ++q) {
        float* base = out + ((size_t)q * n_rows + row0) * K;
        for (int i4 = lane; i4 < run4; i4 += 64) *reinterpret_cast<float4*>(base + 4 * i4) = make_float4(v, v + q, v, v);
    }
}

template <class F> float time_us(F&& f, int reps = 20) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps * 1e3f;
}

int main() {
    const int N = 275167, A = 6;
    {
        const int T = 5;
        float* out; CK(hipMalloc(&out, (size_t)N * T * 128 * 4 + 4096));
        unsigned g = ((N + 3) / 4 + 3) / 4;
        for (int k : {84, 96, 128}) for (int ld : {84, 96, 128}) {
            if (ld < k) continue;
            float t = time_us([&] { hipLaunchKernelGGL(staged_rows, dim3(g), dim3(256), 0, 0, out, N, T, k, ld, 1.f); });
            printf("staged rows, 5 towers, K=%3d LD=%3d : %6.1f us  %.2f TB/s written (%.0f MB)\n", k, ld, t, (double)N * T * k * 4 / t / 1e6, (double)N * T * k * 4 / 1e6);
        }
        for (int R : {4, 8, 16}) for (int wpb : {1, 4}) {
            unsigned gg = ((N + R - 1) / R + wpb - 1) / wpb;
            float t = time_us([&] { hipLaunchKernelGGL(staged_group, dim3(gg), dim3(64 * wpb), 0, 0, out, N, T, 84, R, 1.f); });
            printf("grouped rows, 5 towers, K=84, %2d rows per wave, %d waves per block : %6.1f us  %.2f TB/s\n", R, wpb, t, (double)N * T * 84 * 4 / t / 1e6);
        }
        CK(hipFree(out));
    }
    for (int F : {70}) {
        size_t bytes = (size_t)N * A * F * 4;
        float* out; CK(hipMalloc(&out, bytes + 4096));
        unsigned nb = (N + 3) / 4;
        float t;
        t = time_us([&] { hipLaunchKernelGGL(wave_per_row, dim3(nb), dim3(256), 0, 0, out, N, A, F, 1.f); });
        printf("F=%3d wave/row node-major      : %6.1f us  %.2f TB/s\n", F, t, bytes / t / 1e6);
        for (int R : {2, 4, 8}) {
            unsigned g = ((N + R - 1) / R + 3) / 4;
            t = time_us([&] { hipLaunchKernelGGL(wave_per_rows, dim3(g), dim3(256), 0, 0, out, N, A, F, R, 1.f); });
            printf("F=%3d wave/%d rows node-major   : %6.1f us  %.2f TB/s\n", F, R, t, bytes / t / 1e6);
        }
        if (F % 5 == 0) {
            t = time_us([&] { hipLaunchKernelGGL(wave_per_row_towers, dim3(nb), dim3(256), 0, 0, out, N, A, F, 5, 1.f); });
            printf("F=%3d wave/row tower-major (5)  : %6.1f us  %.2f TB/s\n", F, t, bytes / t / 1e6);
        }
        t = time_us([&] { hipLaunchKernelGGL(flat, dim3(2048), dim3(256), 0, 0, (float2*)out, bytes / 8, 1.f); });
        printf("F=%3d flat float2 stream        : %6.1f us  %.2f TB/s\n", F, t, bytes / t / 1e6);
        t = time_us([&] { hipLaunchKernelGGL(flat, dim3(16384), dim3(256), 0, 0, (float2*)out, bytes / 8, 1.f); });
        printf("F=%3d flat float2 stream (16k)  : %6.1f us  %.2f TB/s\n", F, t, bytes / t / 1e6);
        t = time_us([&] { hipLaunchKernelGGL(block_tile, dim3((N + 15) / 16), dim3(256), 0, 0, out, N, A, F, 1.f); });
        printf("F=%3d workgroup per 16-row tile : %6.1f us  %.2f TB/s\n", F, t, bytes / t / 1e6);
        CK(hipFree(out));
    }
    return 0;
}
