// Feasibility probes for the round-2 kernel plans: what exactly do
//   (1) ds_read_b64_tr_b16  (LDS transpose read, for bf16 MFMA operands whose contraction index is the LDS ROW)
//   (2) global_load_lds_dwordx4 / __builtin_amdgcn_global_load_lds (direct-to-LDS gather, 16 B per lane)
// deliver on gfx950?  Prints the lane -> element mapping; no product code depends on this file.
//   hipcc --offload-arch=gfx950 -O2 -o probe tools/probes/probe_gfx950.hip && ./probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void probe_tr(uint16_t* out /*[64][4]*/, int row_stride_bytes) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = uint16_t(i);
    __syncthreads();
    const int l = threadIdx.x;
    // lane i of a 16-lane group points at the 8-byte piece (row i>>2, columns 4*(i&3) .. +3) of a [4][16] block;
    // group g uses block g (blocks `4 * row_stride` apart)
    const int i = l & 15, g = l >> 4;
    const unsigned addr = unsigned(reinterpret_cast<uintptr_t>(lds)) +
                          unsigned(g * 4 * row_stride_bytes + (i >> 2) * row_stride_bytes + (i & 3) * 8);
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = uint16_t(v & 0xffff);
    out[l * 4 + 1] = uint16_t((v >> 16) & 0xffff);
    out[l * 4 + 2] = uint16_t((v >> 32) & 0xffff);
    out[l * 4 + 3] = uint16_t((v >> 48) & 0xffff);
}

__global__ void probe_glds(const uint32_t* src /*[64*4]*/, uint32_t* out /*[2][64*4]*/) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[512];
    volatile uint32_t* vl = lds;
    const int l = threadIdx.x;
    // every lane fetches 16 bytes from a per-lane (gather) address: lane l reads src row (63 - l)
    const uint32_t* g = src + (63 - l) * 4;
    // (a) the builtin
    for (int i = l; i < 512; i += 64) vl[i] = 0xdeadbeefu;
    __syncthreads();
    __builtin_amdgcn_global_load_lds(g, lds, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = l; i < 256; i += 64) out[i] = vl[i];
    __syncthreads();
    // (b) the instruction itself: M0 = wave-uniform LDS byte address, destination = M0 + lane * 16
    for (int i = l; i < 512; i += 64) vl[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned base = __builtin_amdgcn_readfirstlane(unsigned(reinterpret_cast<uintptr_t>(lds)));
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_waitcnt vmcnt(0)"
                 :: "s"(base), "v"(g) : "memory");
    __syncthreads();
    for (int i = l; i < 256; i += 64) out[256 + i] = vl[i];
}

int main() {
    uint16_t* d;
    (void)hipMalloc(&d, 64 * 4 * 2);
    for (int stride : {32, 80}) {
        hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, 0, d, stride);
        uint16_t h[256];
        (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("ds_read_b64_tr_b16, row stride %d B (element index = byte offset / 2):\n", stride);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d:", l);
            for (int j = 0; j < 4; ++j) {
                const int e = h[l * 4 + j];
                printf("  %4d (row %d col %2d)", e, e / (stride / 2), e % (stride / 2));
            }
            printf("\n");
        }
    }
    uint32_t hs[64 * 4], *ds, *dout;
    for (int i = 0; i < 256; ++i) hs[i] = (i / 4) * 100 + (i % 4);      // row r holds r*100 + 0..3
    (void)hipMalloc(&ds, sizeof hs);
    (void)hipMalloc(&dout, 2 * sizeof hs);
    (void)hipMemcpy(ds, hs, sizeof hs, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_glds, dim3(1), dim3(64), 0, 0, ds, dout);
    uint32_t ho[512];
    hipError_t e = hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    for (int v = 0; v < 2; ++v) {
        printf("global_load_lds 16 B/lane via %s (%s): LDS dwords after lane l gathered src row 63-l:\n",
               v ? "inline asm + M0" : "the builtin", hipGetErrorString(e));
        for (int i = 0; i < 256; i += 16) {
            printf("  lds[%3d..]:", i);
            for (int j = 0; j < 16; j += 4) printf(" %5u", ho[v * 256 + i + j]);
            printf("   (first dword of 4 consecutive 16-byte slots)\n");
        }
    }
    return 0;
}
