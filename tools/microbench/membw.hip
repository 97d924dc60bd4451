// Microbenchmarks that calibrate what the aggregation sweep's access patterns can reach on MI355X:
//   copy      : float4 streaming copy (the guide's 6.29 TB/s reference)
//   write     : wave-per-row streaming writes of ROWB-byte rows with 4/8/16-byte stores per lane
//   gather    : wave-per-task random gathers of 512-byte rows (K rows per wave, U in flight)
// Build: hipcc --offload-arch=gfx950 -O3 membw.hip -o membw ; run: ./membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) b[i] = a[i];
}

// one wave per row of `row_floats` floats; VEC floats per lane per store, segments of 64*VEC floats
template <int VEC>
__global__ void write_rows(float* __restrict__ out, size_t n_rows, int row_floats, float v) {
    size_t row = (size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    int lane = threadIdx.x & 63;
    float* p = out + row * (size_t)row_floats;
    for (int c = lane * VEC; c < row_floats; c += 64 * VEC) {
        if constexpr (VEC == 4) *reinterpret_cast<float4*>(p + c) = make_float4(v, v, v, v);
        else if constexpr (VEC == 2) *reinterpret_cast<float2*>(p + c) = make_float2(v, v);
        else p[c] = v;
    }
}

// same rows, but a persistent grid: each wave loops over rows (grid-stride)
template <int VEC>
__global__ void write_rows_persist(float* __restrict__ out, size_t n_rows, int row_floats, float v) {
    size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    size_t n_waves = (size_t)gridDim.x * (blockDim.x / 64);
    int lane = threadIdx.x & 63;
    for (size_t row = wave; row < n_rows; row += n_waves) {
        float* p = out + row * (size_t)row_floats;
        for (int c = lane * VEC; c < row_floats; c += 64 * VEC) {
            if constexpr (VEC == 4) *reinterpret_cast<float4*>(p + c) = make_float4(v, v, v, v);
            else if constexpr (VEC == 2) *reinterpret_cast<float2*>(p + c) = make_float2(v, v);
            else p[c] = v;
        }
    }
}

// one wave per task: gather K random 512-B rows (float2 per lane), U loads in flight
template <int U>
__global__ void gather_rows(const float* __restrict__ x, const int* __restrict__ idx, size_t n_tasks, int K, float* __restrict__ out) {
    size_t task = (size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (task >= n_tasks) return;
    int lane = threadIdx.x & 63;
    const int* my = idx + task * K;
    float2 acc = make_float2(0.f, 0.f);
    for (int k = 0; k < K; k += U) {
        float2 m[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int s = __builtin_amdgcn_readfirstlane(my[k + u]);
            m[u] = *reinterpret_cast<const float2*>(x + (size_t)s * 128 + lane * 2);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += m[u].x; acc.y += m[u].y; }
    }
    if (acc.x == 123.456f) *reinterpret_cast<float2*>(out + task * 128 + lane * 2) = acc;
}

// launch floor: n waves, each reads two ints (a row pointer pair) and does nothing else
__global__ void launch_floor(const int* __restrict__ ptr, size_t n_waves, float* __restrict__ out) {
    size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    int wi = __builtin_amdgcn_readfirstlane((int)w);
    int a = ptr[wi], b = ptr[wi + 1];
    if (b - a == 123456) out[threadIdx.x] = 1.f;
}

// the same plus a dependent chain of `depth` loads (each address depends on the previous value)
__global__ void chain_floor(const int* __restrict__ ptr, size_t n_waves, int depth, size_t n_ptr, float* __restrict__ out) {
    size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    int v = (int)(w % n_ptr);
    int lane = threadIdx.x & 63;
    for (int d = 0; d < depth; ++d) v = ptr[((size_t)v * 97 + lane) % n_ptr] + (int)(w % n_ptr);
    if (v == -123456) out[threadIdx.x] = 1.f;
}

template <typename F>
float time_ms(F&& f, int reps = 5) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const size_t GB = 1ull << 30;
    // launch floor
    {
        size_t n_ptr = 1 << 20;
        std::vector<int> h(n_ptr + 1);
        for (size_t i = 0; i <= n_ptr; ++i) h[i] = (int)(i % 7);
        int* ptr; CK(hipMalloc(&ptr, (n_ptr + 1) * 4)); CK(hipMemcpy(ptr, h.data(), (n_ptr + 1) * 4, hipMemcpyHostToDevice));
        float* out; CK(hipMalloc(&out, 4096));
        for (size_t n_waves : {(size_t)275167, (size_t)1000000}) {
            for (int bs : {64, 256}) {
                unsigned nb = (unsigned)((n_waves + bs / 64 - 1) / (bs / 64));
                float ms = time_ms([&] { hipLaunchKernelGGL(launch_floor, dim3(nb), dim3(bs), 0, 0, ptr, n_waves, out); }, 20);
                printf("launch floor: %8zu waves, block %3d : %.4f ms  (%.0f waves/us)\n", n_waves, bs, ms, n_waves / ms / 1e3);
            }
            for (int depth : {1, 2, 4, 8}) {
                unsigned nb = (unsigned)n_waves;
                float ms = time_ms([&] { hipLaunchKernelGGL(chain_floor, dim3(nb), dim3(64), 0, 0, ptr, n_waves, depth, n_ptr, out); }, 20);
                printf("  dependent-load chain depth %d, %8zu waves: %.4f ms  (%.0f waves/us)\n", depth, n_waves, ms, n_waves / ms / 1e3);
            }
        }
        CK(hipFree(ptr)); CK(hipFree(out));
    }
    if (getenv("FLOOR_ONLY")) return 0;
    // copy
    {
        size_t n = 4 * GB / 16;
        float4 *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
        CK(hipMemset(a, 1, n * 16));
        float ms = time_ms([&] { hipLaunchKernelGGL(copy4, dim3(256 * 8), dim3(256), 0, 0, a, b, n); });
        printf("copy float4 4GiB->4GiB          : %.3f ms  %.2f TB/s (read+write)\n", ms, 2.0 * n * 16 / ms / 1e9);
        CK(hipFree(a)); CK(hipFree(b));
    }
    // writes
    for (int row_floats : {3072, 1024, 350}) {
        size_t n_rows = (size_t)(24 * GB / 4) / row_floats;
        if (row_floats % 4) n_rows = (size_t)(8 * GB / 4) / row_floats;
        float* out; CK(hipMalloc(&out, n_rows * row_floats * 4));
        size_t bytes = n_rows * (size_t)row_floats * 4;
        unsigned nb = (unsigned)((n_rows + 3) / 4);
        float ms;
        if (row_floats % 4 == 0) {
            ms = time_ms([&] { hipLaunchKernelGGL(write_rows<4>, dim3(nb), dim3(256), 0, 0, out, n_rows, row_floats, 1.f); });
            printf("write rows of %5d floats, 16B/lane: %.3f ms  %.2f TB/s\n", row_floats, ms, bytes / ms / 1e9);
            ms = time_ms([&] { hipLaunchKernelGGL(write_rows_persist<4>, dim3(256 * 8), dim3(256), 0, 0, out, n_rows, row_floats, 1.f); });
            printf("  persistent grid (2048 blocks)      : %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
        }
        if (row_floats % 2 == 0) {
            ms = time_ms([&] { hipLaunchKernelGGL(write_rows<2>, dim3(nb), dim3(256), 0, 0, out, n_rows, row_floats, 1.f); });
            printf("write rows of %5d floats,  8B/lane: %.3f ms  %.2f TB/s\n", row_floats, ms, bytes / ms / 1e9);
            ms = time_ms([&] { hipLaunchKernelGGL(write_rows_persist<2>, dim3(256 * 8), dim3(256), 0, 0, out, n_rows, row_floats, 1.f); });
            printf("  persistent grid (2048 blocks)      : %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
        }
        ms = time_ms([&] { hipLaunchKernelGGL(write_rows<1>, dim3(nb), dim3(256), 0, 0, out, n_rows, row_floats, 1.f); });
        printf("write rows of %5d floats,  4B/lane: %.3f ms  %.2f TB/s\n", row_floats, ms, bytes / ms / 1e9);
        CK(hipFree(out));
    }
    // gathers
    {
        size_t n_src = 10'000'000;   // 5.12 GB of 512-B rows
        float* x; CK(hipMalloc(&x, n_src * 512));
        CK(hipMemset(x, 0, n_src * 512));
        for (int K : {4, 16, 64}) {
            size_t n_tasks = (size_t)64'000'000 / K;
            std::vector<int> h(n_tasks * K);
            unsigned long long s = 88172645463325252ull;
            for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (int)(s % n_src); }
            int* idx; CK(hipMalloc(&idx, h.size() * 4));
            CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
            float* out; CK(hipMalloc(&out, 512));
            unsigned nb = (unsigned)((n_tasks + 3) / 4);
            double bytes = (double)n_tasks * K * 512;
            float ms = time_ms([&] { hipLaunchKernelGGL(gather_rows<2>, dim3(nb), dim3(256), 0, 0, x, idx, n_tasks, K, out); }, 3);
            printf("gather 512B rows K=%2d, 2 in flight : %.3f ms  %.2f TB/s\n", K, ms, bytes / ms / 1e9);
            ms = time_ms([&] { hipLaunchKernelGGL(gather_rows<4>, dim3(nb), dim3(256), 0, 0, x, idx, n_tasks, K, out); }, 3);
            printf("gather 512B rows K=%2d, 4 in flight : %.3f ms  %.2f TB/s\n", K, ms, bytes / ms / 1e9);
            if (K >= 16) {
                ms = time_ms([&] { hipLaunchKernelGGL(gather_rows<8>, dim3(nb), dim3(256), 0, 0, x, idx, n_tasks, K, out); }, 3);
                printf("gather 512B rows K=%2d, 8 in flight : %.3f ms  %.2f TB/s\n", K, ms, bytes / ms / 1e9);
            }
            CK(hipFree(idx)); CK(hipFree(out));
        }
        CK(hipFree(x));
    }
    return 0;
}
