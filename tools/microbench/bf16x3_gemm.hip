// Split-bf16 ("bf16x3") tall-skinny product against the exact-fp32 MFMA kernels of the library.
//   C[M, N] = A[M, K] W[N, K]^T,  fp32 in / fp32 out.  x = hi + mid + lo with three bf16 (8 + 8 + 8 mantissa bits); the six products
//   with i + j <= 2 of (hi, mid, lo) x (hi, mid, lo) on v_mfma_f32_16x16x32_bf16 (fp32 accumulate, each bf16 x bf16 product exact):
//   the dropped terms are < 2^-24 relative per product.  A is split while its tile is staged through LDS, W is split once (a [3][N][Kp]
//   bf16 array).  Baseline: dgn_gemm_forward of libdgn_hip.so (v_mfma_f32_16x16x4_f32) on the same operands.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/microbench/bf16x3_gemm.hip -L dgn_amd -ldgn_hip -Wl,-rpath,$PWD/dgn_amd -o tools/microbench/bf16x3_gemm
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dgn_hip.h"

using f4 = __attribute__((ext_vector_type(4))) float;
using s8 = __attribute__((ext_vector_type(8))) short;       // 8 bf16 = one MFMA operand (4 VGPRs)
using u4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int kTileM = 256, kKC = 32;

// fp32 -> bf16 bits, round to nearest even (finite inputs)
__device__ __forceinline__ unsigned bf16_rne(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// x (two floats) -> packed (hi, mid, lo) pairs; v_cvt_pk_bf16_f32 (round to nearest even); residuals are exact in fp32
typedef float f2v __attribute__((ext_vector_type(2)));
typedef __bf16 b2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{a, b}, b2v)); }
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
    mid = pk_bf16(r0, r1);
    const float q0 = r0 - __uint_as_float(mid << 16), q1 = r1 - __uint_as_float(mid & 0xffff0000u);
    lo = pk_bf16(q0, q1);
}

// W [N, K] fp32 -> planes [3][Np][Kp] bf16 (Np, Kp: padded to 16 / 32, zeros)
// KS > 1: position (32-chunk s KS + c, group g, element e) of a row holds k = 32 KS s + 8 KS g + 8 c + e (v3: a lane streams 8 KS consecutive k)
__global__ void split_w(int N, int K, int Np, int Kp, const float* __restrict__ W, unsigned short* __restrict__ P, int KS) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Np * Kp) return;
    const int n = i / Kp, pos = i - n * Kp;
    const int ch = pos >> 5, g = (pos >> 3) & 3, e = pos & 7, sc = ch / KS, c = ch - sc * KS;
    const int k = 32 * KS * sc + 8 * KS * g + 8 * c + e;
    const float x = (n < N && k < K) ? W[(size_t)n * K + k] : 0.f;
    const unsigned h = bf16_rne(x);
    const float r = x - __uint_as_float(h << 16);
    const unsigned m = bf16_rne(r);
    const float q = r - __uint_as_float(m << 16);
    P[i] = (unsigned short)h; P[(size_t)Np * Kp + i] = (unsigned short)m; P[2 * (size_t)Np * Kp + i] = (unsigned short)bf16_rne(q);
}

// 256 rows x 16 NQ columns per workgroup (4 waves x 64 rows), K in chunks of 32.  LDS: A planes [3][256][32] bf16 (48 KB), W planes
// [3][16 NQ][32] (15 KB at NQ = 5): single-buffered, the next chunk's global loads in flight during the MFMAs.
template <int NQ, int TERMS>
__global__ __launch_bounds__(256) void bf3_gemm(int64_t M, int K, int N, const float* __restrict__ A, int64_t lda, const unsigned short* __restrict__ WP,
                                                int Np, int Kp, float* __restrict__ C, int64_t ldc) {
    __shared__ u4 As[3][kTileM][kKC / 8];          // (row, 8-element group) -> 16 bytes
    __shared__ u4 Bs[3][NQ * 16][kKC / 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * kTileM;
    const int n0 = blockIdx.y * NQ * 16;
    const int lr = tid >> 2, c8 = (tid & 3) * 8;          // staging: thread -> (row lr + 64 j, 8 consecutive k)
    f4 acc[4][NQ];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[rt][q] = f4{0.f, 0.f, 0.f, 0.f};
    f4 ra[4][2];
    u4 rb[3][(NQ * 16 * 4 + 255) / 256];
    constexpr int NB = (NQ * 16 * 4 + 255) / 256;
    const int KB = (K + kKC - 1) / kKC;
    auto fetch = [&](int kc) {
        const int k0 = kc * kKC + c8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = min(row0 + lr + 64 * j, M - 1);
            const float* src = A + r * lda + min(k0, K - 8);          // (K % 8 == 0 in this benchmark; ragged K: clamp + mask)
            ra[j][0] = *reinterpret_cast<const f4*>(src);
            ra[j][1] = *reinterpret_cast<const f4*>(src + 4);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int idx = tid + 256 * j;                          // (column n0 + idx / 4, group idx % 4)
                const int col = min(n0 + (idx >> 2), Np - 1), grp = idx & 3;
                rb[pl][j] = *reinterpret_cast<const u4*>(WP + ((size_t)pl * Np + col) * Kp + kc * kKC + 8 * grp);
            }
    };
    auto commit = [&](int kc) {
        const bool live = kc * kKC + c8 < K;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u4 hi, mid, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x0 = live ? ra[j][e >> 1][2 * (e & 1)] : 0.f, x1 = live ? ra[j][e >> 1][2 * (e & 1) + 1] : 0.f;
                unsigned h, m, l;
                split2(x0, x1, h, m, l);
                hi[e] = h; mid[e] = m; lo[e] = l;
            }
            As[0][lr + 64 * j][tid & 3] = hi;
            As[1][lr + 64 * j][tid & 3] = mid;
            As[2][lr + 64 * j][tid & 3] = lo;
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int idx = tid + 256 * j;
                if (idx < NQ * 16 * 4) Bs[pl][idx >> 2][idx & 3] = rb[pl][j];
            }
    };
    auto mma = [&]() {
        s8 xa[4][3];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) xa[rt][pl] = __builtin_bit_cast(s8, As[pl][64 * wave + 16 * rt + i16][g]);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            s8 wb[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wb[pl] = __builtin_bit_cast(s8, Bs[pl][16 * q + i16][g]);
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                f4 a = acc[rt][q];
                // smallest terms first
                if (TERMS >= 6) {
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[1], xa[rt][1], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[2], xa[rt][0], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[0], xa[rt][2], a, 0, 0, 0);
                }
                if (TERMS >= 3) {
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[1], xa[rt][0], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[0], xa[rt][1], a, 0, 0, 0);
                }
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[0], xa[rt][0], a, 0, 0, 0);
                acc[rt][q] = a;
            }
        }
    };
    fetch(0);
    commit(0);
    __syncthreads();
    for (int kc = 0; kc < KB; ++kc) {
        if (kc + 1 < KB) fetch(kc + 1);
        mma();
        __syncthreads();
        if (kc + 1 < KB) commit(kc + 1);
        __syncthreads();
    }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const int64_t r = row0 + 64 * wave + 16 * rt + i16;
        if (r < M) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int col = n0 + 16 * q + 4 * g;
                if (col + 3 < N) *reinterpret_cast<f4*>(C + r * ldc + col) = acc[rt][q];
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (col + e < N) C[r * ldc + col + e] = acc[rt][q][e];
            }
        }
    }
}

// v2: A never touches LDS -- lane (i16, g) of a wave loads the 8 consecutive k of ITS row of each 16-row tile (two dwordx4: a row's
// four lanes cover one 128-byte line) one chunk ahead, splits them in registers into the three MFMA operands; only the (pre-split) W
// chunk goes through LDS, double-buffered: ONE barrier per 32-k chunk and nothing of A behind it.
template <int NQ, int TERMS>
__global__ __launch_bounds__(256) void bf3_gemm_v2(int64_t M, int K, int N, const float* __restrict__ A, int64_t lda, const unsigned short* __restrict__ WP,
                                                   int Np, int Kp, float* __restrict__ C, int64_t ldc) {
    __shared__ u4 Bs[2][3][NQ * 16][kKC / 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * kTileM + 64 * wave;
    const int n0 = blockIdx.y * NQ * 16;
    f4 acc[4][NQ];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[rt][q] = f4{0.f, 0.f, 0.f, 0.f};
    constexpr int NB = (NQ * 16 * 4 + 255) / 256;
    f4 ra[4][2];
    u4 rb[3][NB];
    const int KB = (K + kKC - 1) / kKC;
    const float* arow[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) arow[rt] = A + min(row0 + 16 * rt + i16, M - 1) * lda;
    auto fetch = [&](int kc) {
        const int k0 = min(kc * kKC + 8 * g, K - 8);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            ra[rt][0] = *reinterpret_cast<const f4*>(arow[rt] + k0);
            ra[rt][1] = *reinterpret_cast<const f4*>(arow[rt] + k0 + 4);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int idx = tid + 256 * j;
                const int col = min(n0 + (idx >> 2), Np - 1), grp = idx & 3;
                rb[pl][j] = *reinterpret_cast<const u4*>(WP + ((size_t)pl * Np + col) * Kp + kc * kKC + 8 * grp);
            }
    };
    auto commit_b = [&](int buf) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int idx = tid + 256 * j;
                if (idx < NQ * 16 * 4) Bs[buf][pl][idx >> 2][idx & 3] = rb[pl][j];
            }
    };
    s8 xa[4][3];
    auto split_a = [&](int kc) {
        const bool live = kc * kKC + 8 * g < K;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            u4 hi, mid, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x0 = live ? ra[rt][e >> 1][2 * (e & 1)] : 0.f, x1 = live ? ra[rt][e >> 1][2 * (e & 1) + 1] : 0.f;
                unsigned h, m, l;
                split2(x0, x1, h, m, l);
                hi[e] = h; mid[e] = m; lo[e] = l;
            }
            xa[rt][0] = __builtin_bit_cast(s8, hi); xa[rt][1] = __builtin_bit_cast(s8, mid); xa[rt][2] = __builtin_bit_cast(s8, lo);
        }
    };
    auto mma = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            s8 wb[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wb[pl] = __builtin_bit_cast(s8, Bs[buf][pl][16 * q + i16][g]);
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                f4 a = acc[rt][q];
                if (TERMS >= 6) {
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[1], xa[rt][1], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[2], xa[rt][0], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[0], xa[rt][2], a, 0, 0, 0);
                }
                if (TERMS >= 3) {
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[1], xa[rt][0], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[0], xa[rt][1], a, 0, 0, 0);
                }
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[0], xa[rt][0], a, 0, 0, 0);
                acc[rt][q] = a;
            }
        }
    };
    fetch(0);
    commit_b(0);
    split_a(0);
    __syncthreads();
    for (int kc = 0; kc < KB; ++kc) {
        if (kc + 1 < KB) fetch(kc + 1);
        mma(kc & 1);
        if (kc + 1 < KB) {
            commit_b((kc + 1) & 1);
            split_a(kc + 1);
        }
        __syncthreads();
    }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const int64_t r = row0 + 16 * rt + i16;
        if (r < M) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int col = n0 + 16 * q + 4 * g;
                if (col + 3 < N) __builtin_nontemporal_store(acc[rt][q], reinterpret_cast<f4*>(C + r * ldc + col));
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (col + e < N) C[r * ldc + col + e] = acc[rt][q][e];
            }
        }
    }
}

// v3: as v2, but a lane streams 8 KS consecutive k of its row per fetch (KS = 2: 64 bytes, the row's four lanes cover 256 contiguous
// bytes; KS = 4: 512) -- the k positions of the MFMA operands are a permutation that W's planes follow (split_w), so nothing moves.
template <int NQ, int TERMS, int KS>
__global__ __launch_bounds__(256) void bf3_gemm_v3(int64_t M, int K, int N, const float* __restrict__ A, int64_t lda, const unsigned short* __restrict__ WP,
                                                   int Np, int Kp, float* __restrict__ C, int64_t ldc) {
    __shared__ u4 Bs[2][KS][3][NQ * 16][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * kTileM + 64 * wave;
    const int n0 = blockIdx.y * NQ * 16;
    f4 acc[4][NQ];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[rt][q] = f4{0.f, 0.f, 0.f, 0.f};
    constexpr int NBT = KS * 3 * NQ * 16 * 4;          // 16-byte pieces of a super-chunk of W
    constexpr int NB = (NBT + 255) / 256;
    f4 ra[4][2 * KS];
    u4 rb[NB];
    const int SB = Kp / (32 * KS);
    const float* arow[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) arow[rt] = A + min(row0 + 16 * rt + i16, M - 1) * lda;
    auto fetch = [&](int sc) {
        const int k0 = min(sc * 32 * KS + 8 * KS * g, K - 8 * KS);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int j = 0; j < 2 * KS; ++j) ra[rt][j] = *reinterpret_cast<const f4*>(arow[rt] + k0 + 4 * j);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int idx = min(tid + 256 * j, NBT - 1);
            const int grp = idx & 3, col = (idx >> 2) % (NQ * 16), pl = (idx >> 2) / (NQ * 16) % 3, c = (idx >> 2) / (NQ * 16 * 3);
            rb[j] = *reinterpret_cast<const u4*>(WP + ((size_t)pl * Np + min(n0 + col, Np - 1)) * Kp + (sc * KS + c) * 32 + 8 * grp);
        }
    };
    auto commit_b = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int idx = tid + 256 * j;
            if (idx < NBT) (&Bs[buf][0][0][0][0])[idx] = rb[j];
        }
    };
    s8 xa[KS][4][3];
    auto split_a = [&](int sc) {
        const bool live = sc * 32 * KS + 8 * KS * g < K;
#pragma unroll
        for (int c = 0; c < KS; ++c)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                u4 hi, mid, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f4 v = ra[rt][2 * c + (e >> 1)];
                    const float x0 = live ? v[2 * (e & 1)] : 0.f, x1 = live ? v[2 * (e & 1) + 1] : 0.f;
                    unsigned h, m, l;
                    split2(x0, x1, h, m, l);
                    hi[e] = h; mid[e] = m; lo[e] = l;
                }
                xa[c][rt][0] = __builtin_bit_cast(s8, hi); xa[c][rt][1] = __builtin_bit_cast(s8, mid); xa[c][rt][2] = __builtin_bit_cast(s8, lo);
            }
    };
    auto mma = [&](int buf) {
#pragma unroll
        for (int c = 0; c < KS; ++c)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                s8 wb[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wb[pl] = __builtin_bit_cast(s8, Bs[buf][c][pl][16 * q + i16][g]);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    f4 a = acc[rt][q];
                    if (TERMS >= 6) {
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[1], xa[c][rt][1], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[2], xa[c][rt][0], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[0], xa[c][rt][2], a, 0, 0, 0);
                    }
                    if (TERMS >= 3) {
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[1], xa[c][rt][0], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[0], xa[c][rt][1], a, 0, 0, 0);
                    }
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[0], xa[c][rt][0], a, 0, 0, 0);
                    acc[rt][q] = a;
                }
            }
    };
    fetch(0);
    commit_b(0);
    split_a(0);
    __syncthreads();
    for (int sc = 0; sc < SB; ++sc) {
        if (sc + 1 < SB) fetch(sc + 1);
        mma(sc & 1);
        if (sc + 1 < SB) {
            commit_b((sc + 1) & 1);
            split_a(sc + 1);
        }
        __syncthreads();
    }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const int64_t r = row0 + 16 * rt + i16;
        if (r < M) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int col = n0 + 16 * q + 4 * g;
                if (col + 3 < N) __builtin_nontemporal_store(acc[rt][q], reinterpret_cast<f4*>(C + r * ldc + col));
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (col + e < N) C[r * ldc + col + e] = acc[rt][q][e];
            }
        }
    }
}

// v4: 128-row tiles (a wave owns 32 rows: RT = 2), so that four workgroups = sixteen waves stay resident per CU, and the A rows TWO
// chunks ahead (two register sets, the k loop unrolled by two): twice the bytes in flight of v2.  W's planes as in v2.
template <int NQ, int TERMS>
__global__ __launch_bounds__(256) void bf3_gemm_v4(int64_t M, int K, int N, const float* __restrict__ A, int64_t lda, const unsigned short* __restrict__ WP,
                                                   int Np, int Kp, float* __restrict__ C, int64_t ldc) {
    constexpr int RT = 2;
    __shared__ u4 Bs[2][3][NQ * 16][kKC / 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * (64 * RT) + 16 * RT * wave;
    const int n0 = blockIdx.y * NQ * 16;
    f4 acc[RT][NQ];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[rt][q] = f4{0.f, 0.f, 0.f, 0.f};
    constexpr int NB = (NQ * 16 * 4 + 255) / 256;
    const int KB = (K + kKC - 1) / kKC;
    const float* arow[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) arow[rt] = A + min(row0 + 16 * rt + i16, M - 1) * lda;
    struct ASet { f4 v[RT][2]; };
    auto fetch_a = [&](ASet& a, int kc) {
        const int k0 = min(kc * kKC + 8 * g, K - 8);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            a.v[rt][0] = *reinterpret_cast<const f4*>(arow[rt] + k0);
            a.v[rt][1] = *reinterpret_cast<const f4*>(arow[rt] + k0 + 4);
        }
    };
    u4 rb[3][NB];
    auto fetch_b = [&](int kc) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int idx = tid + 256 * j;
                const int col = min(n0 + (idx >> 2), Np - 1), grp = idx & 3;
                rb[pl][j] = *reinterpret_cast<const u4*>(WP + ((size_t)pl * Np + col) * Kp + kc * kKC + 8 * grp);
            }
    };
    auto commit_b = [&](int buf) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int idx = tid + 256 * j;
                if (idx < NQ * 16 * 4) Bs[buf][pl][idx >> 2][idx & 3] = rb[pl][j];
            }
    };
    s8 xa[RT][3];
    auto split_a = [&](const ASet& a, int kc) {
        const bool live = kc * kKC + 8 * g < K;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            u4 hi, mid, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x0 = live ? a.v[rt][e >> 1][2 * (e & 1)] : 0.f, x1 = live ? a.v[rt][e >> 1][2 * (e & 1) + 1] : 0.f;
                unsigned h, m, l;
                split2(x0, x1, h, m, l);
                hi[e] = h; mid[e] = m; lo[e] = l;
            }
            xa[rt][0] = __builtin_bit_cast(s8, hi); xa[rt][1] = __builtin_bit_cast(s8, mid); xa[rt][2] = __builtin_bit_cast(s8, lo);
        }
    };
    auto mma = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            s8 wb[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wb[pl] = __builtin_bit_cast(s8, Bs[buf][pl][16 * q + i16][g]);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                f4 a = acc[rt][q];
                if (TERMS >= 6) {
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[1], xa[rt][1], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[2], xa[rt][0], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[0], xa[rt][2], a, 0, 0, 0);
                }
                if (TERMS >= 3) {
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[1], xa[rt][0], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[0], xa[rt][1], a, 0, 0, 0);
                }
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[0], xa[rt][0], a, 0, 0, 0);
                acc[rt][q] = a;
            }
        }
    };
    ASet a0, a1;
    fetch_a(a0, 0);
    fetch_b(0);
    if (KB > 1) fetch_a(a1, 1);
    commit_b(0);
    split_a(a0, 0);
    __syncthreads();
    for (int kc = 0; kc < KB; kc += 2) {
        // chunk kc: operands split, W in buffer 0; set a1 holds chunk kc + 1 (in flight); request kc + 2 into a0
        if (kc + 2 < KB) fetch_a(a0, kc + 2);
        if (kc + 1 < KB) fetch_b(kc + 1);
        mma(0);
        if (kc + 1 >= KB) break;
        commit_b(1);
        split_a(a1, kc + 1);
        __syncthreads();
        if (kc + 3 < KB) fetch_a(a1, kc + 3);
        if (kc + 2 < KB) fetch_b(kc + 2);
        mma(1);
        if (kc + 2 < KB) {
            commit_b(0);
            split_a(a0, kc + 2);
        }
        __syncthreads();
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int64_t r = row0 + 16 * rt + i16;
        if (r < M) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int col = n0 + 16 * q + 4 * g;
                if (col + 3 < N) __builtin_nontemporal_store(acc[rt][q], reinterpret_cast<f4*>(C + r * ldc + col));
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (col + e < N) C[r * ldc + col + e] = acc[rt][q][e];
            }
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NQ, int TERMS>
float run_bf3(int64_t M, int K, int N, const float* A, const unsigned short* WP, int Np, int Kp, float* C, int reps) {
    dim3 grid((unsigned)((M + kTileM - 1) / kTileM), (unsigned)((N + NQ * 16 - 1) / (NQ * 16)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    static const bool v1 = getenv("BF3_V1") != nullptr;
    auto launch = [&]() {
        static const int ks = getenv("BF3_KS") ? atoi(getenv("BF3_KS")) : 1;       // 1: kernel v2 (the table of profiles/r04_bf16x3_gemm.txt), 2: v3
        static const bool v4 = getenv("BF3_V4") != nullptr;
        if (v4) hipLaunchKernelGGL((bf3_gemm_v4<NQ, TERMS>), dim3((unsigned)((M + 127) / 128), grid.y), dim3(256), 0, 0, M, K, N, A, (int64_t)K, WP, Np, Kp, C, (int64_t)N);
        else if (v1) hipLaunchKernelGGL((bf3_gemm<NQ, TERMS>), grid, dim3(256), 0, 0, M, K, N, A, (int64_t)K, WP, Np, Kp, C, (int64_t)N);
        else if (ks == 1) hipLaunchKernelGGL((bf3_gemm_v2<NQ, TERMS>), grid, dim3(256), 0, 0, M, K, N, A, (int64_t)K, WP, Np, Kp, C, (int64_t)N);
        else hipLaunchKernelGGL((bf3_gemm_v3<NQ, TERMS, 2>), grid, dim3(256), 0, 0, M, K, N, A, (int64_t)K, WP, Np, Kp, C, (int64_t)N);      // (KS = 4 spilled: measured once, not kept)
    };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    struct Shape { const char* name; int64_t M; int K, N; };
    const Shape shapes[] = {{"c2c posttrans fwd (agg -> 70)", 275167, 432, 80}, {"c2c input grad (80 -> 432)", 275167, 80, 432},
                            {"c1 posttrans fwd (152 -> 75)", 275167, 152, 76}, {"c4_mega fwd (350 -> 70)", 1049000, 352, 72},
                            {"c5 layer, one scaler (1024 -> 128), 1M rows", 1000000, 1024, 128}};
    const int reps = 10;
    printf("%-46s %10s %10s %10s %10s | %9s %9s %9s | max rel err vs fp64 (256 sampled rows): fp32-mfma, bf16x6, bf16x3, bf16x1\n", "shape", "fp32 ms", "x6 ms",
           "x3 ms", "x1 ms", "fp32 TF", "x6 TF", "speedup");
    for (const Shape& s : shapes) {
        const int64_t M = s.M; const int K = s.K, N = s.N;
        const int ks = getenv("BF3_KS") ? atoi(getenv("BF3_KS")) : 1;       // 1: kernel v2 (the table of profiles/r04_bf16x3_gemm.txt), 2: v3
        const int Np = (N + 15) / 16 * 16, Kp = (K + 32 * ks - 1) / (32 * ks) * (32 * ks);
        std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
        unsigned h = 12345u;
        auto rnd = [&]() { h = h * 1664525u + 1013904223u; return ((h >> 8) * (1.0f / 8388608.0f)) - 1.0f; };
        for (auto& v : hA) v = rnd() * 3.0f;
        for (auto& v : hW) v = rnd() / std::sqrt((float)K);
        float *dA, *dW, *dC0, *dC1; unsigned short* dWP;
        CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&dC0, (size_t)M * N * 4)); CK(hipMalloc(&dC1, (size_t)M * N * 4));
        CK(hipMalloc(&dWP, (size_t)3 * Np * Kp * 2));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(split_w, dim3((Np * Kp + 255) / 256), dim3(256), 0, 0, N, K, Np, Kp, dW, dWP, ks);
        // baseline: the library's exact-fp32 MFMA product
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) if (dgn_gemm_forward(M, K, N, dA, K, dW, K, 0, nullptr, dC0, N, nullptr)) { printf("dgn_gemm_forward: %s\n", dgn_last_error()); return 1; }
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) dgn_gemm_forward(M, K, N, dA, K, dW, K, 0, nullptr, dC0, N, nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms32; CK(hipEventElapsedTime(&ms32, e0, e1)); ms32 /= reps;
        auto go = [&](int terms) -> float {
            const int tiles5 = (N + 79) / 80 * 80 - N, tiles4 = (N + 63) / 64 * 64 - N, tiles7 = (N + 111) / 112 * 112 - N;
            int nq = 5; int pad = tiles5;
            if (tiles4 < pad) { nq = 4; pad = tiles4; }
            if (tiles7 < pad) { nq = 7; pad = tiles7; }
            if (N == 128) nq = 4;
#define RUN(NQV) (terms == 6 ? run_bf3<NQV, 6>(M, K, N, dA, dWP, Np, Kp, dC1, reps) : terms == 3 ? run_bf3<NQV, 3>(M, K, N, dA, dWP, Np, Kp, dC1, reps) : run_bf3<NQV, 1>(M, K, N, dA, dWP, Np, Kp, dC1, reps))
            return nq == 4 ? RUN(4) : nq == 5 ? RUN(5) : RUN(7);
        };
        // errors against fp64 on sampled rows
        std::vector<float> c0((size_t)M * N), c1((size_t)M * N);
        auto err = [&](const std::vector<float>& c) {
            double worst = 0.0;
            for (int t = 0; t < 256; ++t) {
                const int64_t r = (int64_t)((double)t / 256 * (M - 1));
                double scale = 0.0; std::vector<double> ref(N);
                for (int n = 0; n < N; ++n) { double a = 0; for (int k = 0; k < K; ++k) a += (double)hA[r * K + k] * hW[(size_t)n * K + k]; ref[n] = a; scale = std::max(scale, std::fabs(a)); }
                for (int n = 0; n < N; ++n) worst = std::max(worst, std::fabs((double)c[r * N + n] - ref[n]) / scale);
            }
            return worst;
        };
        CK(hipMemcpy(c0.data(), dC0, c0.size() * 4, hipMemcpyDeviceToHost));
        const double e32 = err(c0);
        const float ms6 = go(6); CK(hipDeviceSynchronize()); CK(hipMemcpy(c1.data(), dC1, c1.size() * 4, hipMemcpyDeviceToHost)); const double e6 = err(c1);
        const float ms3 = go(3); CK(hipDeviceSynchronize()); CK(hipMemcpy(c1.data(), dC1, c1.size() * 4, hipMemcpyDeviceToHost)); const double e3 = err(c1);
        const float ms1 = go(1); CK(hipDeviceSynchronize()); CK(hipMemcpy(c1.data(), dC1, c1.size() * 4, hipMemcpyDeviceToHost)); const double e1x = err(c1);
        const double fl = 2.0 * M * K * N;
        printf("%-46s %10.4f %10.4f %10.4f %10.4f | %9.1f %9.1f %9.2f | %.2e %.2e %.2e %.2e\n", s.name, ms32, ms6, ms3, ms1, fl / ms32 / 1e9, fl / ms6 / 1e9, ms32 / ms6, e32, e6, e3, e1x);
        hipFree(dA); hipFree(dW); hipFree(dC0); hipFree(dC1); hipFree(dWP);
    }
    return 0;
}
