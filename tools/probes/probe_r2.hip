// Round-2 hardware probes (build: hipcc --offload-arch=gfx950 -O3 probe_r2.hip -o probe_r2):
//  P1  v_mfma_f32_16x16x32_bf16 operand / result lane mapping against a host matmul (asymmetric B);
//  P2  ds_add_f32 (atomicAdd on LDS) is a plain fp32 RNE add: bit-compare with a register fp32 chain,
//      including subnormal partial sums;
//  P3  cost of adding a 16 x 48 MFMA result block into an LDS fp32 tile at scattered rows:
//      ds_add_f32 vs ds_read + v_add + ds_write, cycles per wave with 4 waves per workgroup.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void p1_kernel(const float* A, const float* B, float* C) {   // A[16][32], B[32][16] row-major fp32 (bf16-exact values)
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)A[(l & 15) * 32 + 8 * (l >> 4) + e];
        b[e] = (__bf16)B[(8 * (l >> 4) + e) * 16 + (l & 15)];
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}

__global__ void p2_kernel(const float* v, int n, float* out_lds, float* out_reg) {
    __shared__ float cell[256];
    const int t = threadIdx.x;
    cell[t] = 0.f;
    __syncthreads();
    float s = 0.f;
    for (int j = 0; j < n; ++j) {
        const float x = v[j * 256 + t];
        atomicAdd(&cell[t], x);
        s += x;
    }
    __syncthreads();
    out_lds[t] = cell[t];
    out_reg[t] = s;
}

template <int MODE>
__global__ __launch_bounds__(256) void p3_kernel(const int* rows, int iters, float* sink, long long* cycles) {
    __shared__ float tile[128 * 100];
    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    for (int i = t; i < 128 * 100; i += 256) tile[i] = 0.f;
    __syncthreads();
    const int cg = w >> 1, ph = w & 1;
    float acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = float(i + l);
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        // 3 column blocks x 4 result registers: row = rows[...], col = cg*48 + b*16 + (l&15)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rows[(it * 32 + ph * 16 + 4 * (l >> 4) + r) & 4095];
                float* p = &tile[row * 100 + cg * 48 + b * 16 + (l & 15)];
                if (MODE == 0) atomicAdd(p, acc[b * 4 + r]);
                else *p = *p + acc[b * 4 + r];
            }
        }
        __syncthreads();
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = t; i < 128 * 100; i += 256) s += tile[i];
    sink[blockIdx.x * 256 + t] = s;
    if (t == 0) cycles[blockIdx.x] = t1 - t0;
}

static float bf16_round(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000u;
    memcpy(&x, &u, 4);
    return x;
}

int main() {
    // ---- P1
    {
        std::vector<float> A(16 * 32), B(32 * 16), C(16 * 16), R(16 * 16, 0.f);
        for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = bf16_round(float((i * 7 + k * 3) % 11) - 5.f);
        for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = bf16_round(float((k * 5 + j * j) % 13) - 6.f);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += A[i * 32 + k] * B[k * 16 + j]; R[i * 16 + j] = s; }
        float *dA, *dB, *dC;
        CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, C.size() * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(p1_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += (C[i] != R[i]);
        printf("P1 mfma_f32_16x16x32_bf16 mapping (A row=l&15,k=8(l>>4)+e; B k=8(l>>4)+e,col=l&15; C row=4(l>>4)+r,col=l&15): %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    }
    // ---- P2
    {
        const int n = 512;
        std::vector<float> v(n * 256);
        srand(1);
        for (size_t i = 0; i < v.size(); ++i) {
            float x = (float(rand()) / RAND_MAX - 0.5f) * powf(10.f, float(rand() % 12) - 6.f);
            if (i % 256 < 32) x *= 1e-38f;                 // columns 0..31: subnormal-range sums
            v[i] = x;
        }
        float *dv, *o1, *o2;
        CK(hipMalloc(&dv, v.size() * 4)); CK(hipMalloc(&o1, 1024)); CK(hipMalloc(&o2, 1024));
        CK(hipMemcpy(dv, v.data(), v.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(p2_kernel, dim3(1), dim3(256), 0, 0, dv, n, o1, o2);
        std::vector<float> a(256), b(256);
        CK(hipMemcpy(a.data(), o1, 1024, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), o2, 1024, hipMemcpyDeviceToHost));
        int bad = 0, bad_sub = 0;
        for (int i = 0; i < 256; ++i) { if (memcmp(&a[i], &b[i], 4)) { ++bad; if (i < 32) ++bad_sub; } }
        printf("P2 ds_add_f32 vs register fp32 chain: %d of 256 cells differ (%d of them in the 32 subnormal-range cells); sample lds=%g reg=%g\n", bad, bad_sub, a[0], b[0]);
    }
    // ---- P3
    {
        std::vector<int> rows(4096);
        srand(2);
        for (int i = 0; i < 4096; ++i) rows[i] = rand() % 128;
        int* dr; float* sink; long long* cyc;
        const int blocks = 512, iters = 200;
        CK(hipMalloc(&dr, 4096 * 4)); CK(hipMalloc(&sink, blocks * 256 * 4)); CK(hipMalloc(&cyc, blocks * 8));
        CK(hipMemcpy(dr, rows.data(), 4096 * 4, hipMemcpyHostToDevice));
        for (int mode = 0; mode < 2; ++mode) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(p3_kernel<0>, dim3(blocks), dim3(256), 0, 0, dr, iters, sink, cyc);
                else hipLaunchKernelGGL(p3_kernel<1>, dim3(blocks), dim3(256), 0, 0, dr, iters, sink, cyc);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
            }
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<long long> c(blocks);
            CK(hipMemcpy(c.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
            double avg = 0;
            for (auto x : c) avg += double(x);
            avg /= blocks;
            printf("P3 %s: %.1f clock64 ticks per 32-pair block update (12 elements/lane, 4 waves, 2 WG/CU), kernel %.3f ms\n",
                   mode == 0 ? "ds_add_f32      " : "read+add+write  ", avg / iters, ms);
        }
    }
    return 0;
}
