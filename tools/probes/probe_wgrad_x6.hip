// Round-2 prototype (NOT product code): the weight-gradient contraction  W[ci][co] = sum_p A[p][ci] * G[p][co]
// on the bf16 MFMA with the three-way operand split of the forward kernel ("bf16x6"), operands staged
// as ROW-MAJOR [pair][channel] bf16 planes in LDS and fetched as MFMA fragments with the gfx950 LDS
// transpose read (ds_read_b64_tr_b16; lane mapping measured by probe_gfx950.hip).  Dense rows (no gather)
// -- this isolates the staging + MFMA pipeline that the product's fp32-MFMA weight-gradient kernel
// spends its time in.  Self-checking against a float64 host reference; prints time and TFLOP/s next to
// an fp32-MFMA version of the same loop.
//   hipcc --offload-arch=gfx950 -O3 -o probe_wgrad tools/probes/probe_wgrad_x6.hip && ./probe_wgrad
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int C = 96;            // input and output channels of the layer (3 x 3 MFMA tiles)
constexpr int RB = 32;           // pairs per stage
constexpr int LD = C + 8;        // bf16 row pitch of the LDS planes (208 B: 16-byte aligned rows)

__device__ __forceinline__ void split3(float v, __bf16& h1, __bf16& h2, __bf16& h3) {
    h1 = (__bf16)v;
    const float r1 = v - (float)h1;
    h2 = (__bf16)r1;
    h3 = (__bf16)(r1 - (float)h2);
}

// LDS byte address of the 8-byte piece this lane contributes to a transpose read of the [4 pairs][16 channels]
// block at (pair k0, channel c16) of a row-major [pair][channel] plane: lane i of each 16-lane group points at
// (row i >> 2, channels 4 (i & 3) .. + 3) and RECEIVES channel i, pairs k0 .. k0 + 3.
__device__ __forceinline__ unsigned tr_addr(const __bf16* plane, int k0, int c16, int lane) {
    const int i = lane & 15;
    return unsigned(reinterpret_cast<uintptr_t>(plane + (k0 + (i >> 2)) * LD + c16 + 4 * (i & 3)));
}

union Frag { unsigned long long u[2]; bf16x8 v; };

// the six operand fragments of one (tile, k-step): 12 transpose reads issued back to back, one wait
__device__ __forceinline__ void load_frags(const __bf16* ap, const __bf16* gp, int k0, int ca, int cb, int lane,
                                           Frag (&a)[3], Frag (&b)[3]) {
    const unsigned aa = tr_addr(ap, k0, ca, lane), ga = tr_addr(gp, k0, cb, lane);
    constexpr unsigned PL = RB * LD * 2;        // bytes between planes
    constexpr unsigned R4 = 4 * LD * 2;         // bytes between pairs k0 and k0 + 4
    asm volatile(
        "ds_read_b64_tr_b16 %0, %12\n\t"
        "ds_read_b64_tr_b16 %1, %12 offset:%14\n\t"
        "ds_read_b64_tr_b16 %2, %12 offset:%15\n\t"
        "ds_read_b64_tr_b16 %3, %12 offset:%16\n\t"
        "ds_read_b64_tr_b16 %4, %12 offset:%17\n\t"
        "ds_read_b64_tr_b16 %5, %12 offset:%18\n\t"
        "ds_read_b64_tr_b16 %6, %13\n\t"
        "ds_read_b64_tr_b16 %7, %13 offset:%14\n\t"
        "ds_read_b64_tr_b16 %8, %13 offset:%15\n\t"
        "ds_read_b64_tr_b16 %9, %13 offset:%16\n\t"
        "ds_read_b64_tr_b16 %10, %13 offset:%17\n\t"
        "ds_read_b64_tr_b16 %11, %13 offset:%18\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(a[0].u[0]), "=&v"(a[0].u[1]), "=&v"(a[1].u[0]), "=&v"(a[1].u[1]), "=&v"(a[2].u[0]), "=&v"(a[2].u[1]),
          "=&v"(b[0].u[0]), "=&v"(b[0].u[1]), "=&v"(b[1].u[0]), "=&v"(b[1].u[1]), "=&v"(b[2].u[0]), "=&v"(b[2].u[1])
        : "v"(aa), "v"(ga), "n"(R4), "n"(PL), "n"(PL + R4), "n"(2 * PL), "n"(2 * PL + R4)
        : "memory");
}

template <bool X6>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const float* __restrict__ A, const float* __restrict__ G,
                                                      int pairs_per_block, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) __bf16 Ap[3][RB][LD];
    __shared__ __attribute__((aligned(16))) __bf16 Gp[3][RB][LD];
    __shared__ __attribute__((aligned(16))) float Af[RB][C + 4];
    __shared__ __attribute__((aligned(16))) float Gf[RB][C + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t p_begin = int64_t(blockIdx.x) * pairs_per_block;
    f32x16 acc[3];
    for (int t = 0; t < 3; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // software pipeline: the rows of stage s+1 are fetched into registers before the MFMAs of stage s and
    // converted / written to LDS after them (what the product kernels do)
    float4 ra[3], rg[3];
    auto fetch = [&](int p0) {
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            const int f = tid + h * 256;
            const int pr = f / (C / 4), q = f - pr * (C / 4);
            ra[h] = *reinterpret_cast<const float4*>(A + (p_begin + p0 + pr) * C + 4 * q);
            rg[h] = *reinterpret_cast<const float4*>(G + (p_begin + p0 + pr) * C + 4 * q);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            const int f = tid + h * 256;
            const int pr = f / (C / 4), q = f - pr * (C / 4);
            const float4 a = ra[h], g = rg[h];
            if (X6) {
                const float av[4] = {a.x, a.y, a.z, a.w}, gv[4] = {g.x, g.y, g.z, g.w};
                bf16x4 a1, a2, a3, g1, g2, g3;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    __bf16 x, y, z;
                    split3(av[e], x, y, z); a1[e] = x; a2[e] = y; a3[e] = z;
                    split3(gv[e], x, y, z); g1[e] = x; g2[e] = y; g3[e] = z;
                }
                *reinterpret_cast<bf16x4*>(&Ap[0][pr][4 * q]) = a1;
                *reinterpret_cast<bf16x4*>(&Ap[1][pr][4 * q]) = a2;
                *reinterpret_cast<bf16x4*>(&Ap[2][pr][4 * q]) = a3;
                *reinterpret_cast<bf16x4*>(&Gp[0][pr][4 * q]) = g1;
                *reinterpret_cast<bf16x4*>(&Gp[1][pr][4 * q]) = g2;
                *reinterpret_cast<bf16x4*>(&Gp[2][pr][4 * q]) = g3;
            } else {
                *reinterpret_cast<float4*>(&Af[pr][4 * q]) = a;
                *reinterpret_cast<float4*>(&Gf[pr][4 * q]) = g;
            }
        }
    };
    fetch(0);
    for (int p0 = 0; p0 < pairs_per_block; p0 += RB) {
        stash();
        __syncthreads();
        if (p0 + RB < pairs_per_block) fetch(p0 + RB);
        // ---- 9 tiles of 32 x 32 dealt to the 4 waves (3 / 2 / 2 / 2)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int tile = wave + 4 * t;
            if (tile < 9) {
                const int ti = tile / 3, tj = tile - 3 * ti;
                if (X6) {
                    // operand row (channel) of this lane = lane & 31; its 16-lane group covers channels
                    // 16 * ((lane >> 4) & 1) .. + 15 of the tile and the k-half lane >> 5
                    const int ca = ti * 32 + 16 * ((lane >> 4) & 1), cb = tj * 32 + 16 * ((lane >> 4) & 1);
                    const int kh = 8 * (lane >> 5);
#pragma unroll
                    for (int ks = 0; ks < RB / 16; ++ks) {
                        const int k0 = ks * 16 + kh;
                        Frag fa[3], fb[3];
                        load_frags(&Ap[0][0][0], &Gp[0][0][0], k0, ca, cb, lane, fa, fb);
                        const bf16x8 a1 = fa[0].v, a2 = fa[1].v, a3 = fa[2].v;
                        const bf16x8 b1 = fb[0].v, b2 = fb[1].v, b3 = fb[2].v;
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[t], 0, 0, 0);
                    }
                } else {
                    const int ca = ti * 32 + (lane & 31), cb = tj * 32 + (lane & 31), kh = lane >> 5;
#pragma unroll
                    for (int kk = 0; kk < RB / 2; ++kk)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Af[2 * kk + kh][ca], Gf[2 * kk + kh][cb], acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    float* d = partial + int64_t(blockIdx.x) * C * C;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int tile = wave + 4 * t;
        if (tile < 9) {
            const int ti = tile / 3, tj = tile - 3 * ti;
            const int co = tj * 32 + (lane & 31);
            for (int r = 0; r < 16; ++r) {
                const int ci = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                d[ci * C + co] = acc[t][r];
            }
        }
    }
}

int main() {
    const int blocks = 512, ppb = 1024;                 // 524 288 pairs ~ the level-0 3^3 map of S100k
    const int64_t P = int64_t(blocks) * ppb;
    std::vector<float> hA(P * C), hG(P * C);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return float(int32_t(s >> 8) - (1 << 23)) / float(1 << 23); };
    for (auto& v : hA) v = rnd();
    for (auto& v : hG) v = rnd() * 0.01f;
    float *dA, *dG, *dP;
    (void)hipMalloc(&dA, P * C * 4); (void)hipMalloc(&dG, P * C * 4); (void)hipMalloc(&dP, size_t(blocks) * C * C * 4);
    (void)hipMemcpy(dA, hA.data(), P * C * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dG, hG.data(), P * C * 4, hipMemcpyHostToDevice);
    // float64 reference of a sample of entries
    const int samples[6][2] = {{0, 0}, {5, 77}, {31, 32}, {64, 95}, {95, 0}, {47, 48}};
    double ref[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t p = 0; p < P; ++p)
        for (int q = 0; q < 6; ++q) ref[q] += double(hA[p * C + samples[q][0]]) * double(hG[p * C + samples[q][1]]);
    std::vector<float> hP(size_t(blocks) * C * C);
    for (int x6 = 0; x6 < 2; ++x6) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            if (rep == 1) (void)hipEventRecord(e0, 0);
            if (x6) hipLaunchKernelGGL(wgrad_kernel<true>, dim3(blocks), dim3(256), 0, 0, dA, dG, ppb, dP);
            else hipLaunchKernelGGL(wgrad_kernel<false>, dim3(blocks), dim3(256), 0, 0, dA, dG, ppb, dP);
        }
        (void)hipEventRecord(e1, 0);
        hipError_t err = hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        ms /= 2;
        (void)hipMemcpy(hP.data(), dP, hP.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0;
        for (int q = 0; q < 6; ++q) {
            double got = 0;
            for (int b = 0; b < blocks; ++b) got += double(hP[size_t(b) * C * C + samples[q][0] * C + samples[q][1]]);
            worst = fmax(worst, fabs(got - ref[q]));
            scale = fmax(scale, fabs(ref[q]));
        }
        printf("%s: %s  %.1f us  = %.1f TFLOP/s exact   max|d| = %.3e  (max|ref| = %.3e, rel %.2e)\n",
               x6 ? "bf16x6 + tr_b16" : "fp32 MFMA      ", hipGetErrorString(err), ms * 1e3,
               2.0 * P * C * C / (ms * 1e-3) / 1e12, worst, scale, worst / scale);
    }
    return 0;
}
