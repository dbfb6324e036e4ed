// How much does row-gather locality buy on MI355X?  Gathers N rows of 384 B (96 floats, the 96-channel feature
// rows of the U-Net's level 0) from a 39 MB table, 16 B per lane, in three index orders:
//   random (hash order, what the weight-gradient kernel sees today), window-local (random inside windows of W
//   rows: the same rows are fetched again by later windows' neighbours -> L2-sized working set), sequential.
// Every row is read `reuse` times in total (5 = the average number of 3^3 offsets that touch an input row).
// Prints GB/s per order.  Not product code (round-2 planning).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <random>
#include <vector>

constexpr int ROW = 96;                       // floats per row

__global__ void gather_kernel(const float* __restrict__ tab, const int* __restrict__ idx, int64_t n_idx,
                              float* __restrict__ sink) {
    // 24 lanes per row (24 x 16 B = 384 B); 256 threads = 10 rows + 16 idle lanes per pass
    const int t = threadIdx.x;
    const int r_in_blk = t / 24, j = t % 24;
    float acc = 0.f;
    for (int64_t base = int64_t(blockIdx.x) * 10; base < n_idx; base += int64_t(gridDim.x) * 10) {
        const int64_t i = base + r_in_blk;
        if (r_in_blk < 10 && i < n_idx) {
            const float4 v = *reinterpret_cast<const float4*>(tab + int64_t(idx[i]) * ROW + 4 * j);
            acc += v.x + v.y + v.z + v.w;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;      // keep the loads alive
}

int main() {
    const int n_rows = 100999, reuse = 5;
    const int64_t n_idx = int64_t(n_rows) * reuse;
    std::vector<float> tab(size_t(n_rows) * ROW, 1.0f);
    float *dt, *ds; int* di;
    (void)hipMalloc(&dt, tab.size() * 4); (void)hipMalloc(&ds, 4); (void)hipMalloc(&di, n_idx * 4);
    (void)hipMemcpy(dt, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
    std::mt19937 rng(1);
    auto run = [&](const char* name, std::vector<int>& idx) {
        (void)hipMemcpy(di, idx.data(), n_idx * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 6; ++rep) {
            if (rep == 1) (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(gather_kernel, dim3(2048), dim3(256), 0, 0, dt, di, n_idx, ds);
        }
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-34s %7.1f us  %7.1f GB/s\n", name, ms * 1e3, double(n_idx) * ROW * 4 / (ms * 1e-3) / 1e9);
    };
    std::vector<int> idx(n_idx);
    for (int64_t i = 0; i < n_idx; ++i) idx[i] = int(i % n_rows);
    run("sequential (x5 passes)", idx);
    std::shuffle(idx.begin(), idx.end(), rng);
    run("random (hash order)", idx);
    for (int W : {2048, 8192, 32768}) {       // 0.77 MB / 3 MB / 12 MB windows, each row `reuse` times inside its window
        std::vector<int> w; w.reserve(n_idx);
        for (int b = 0; b < n_rows; b += W) {
            std::vector<int> blk;
            for (int r = b; r < std::min(n_rows, b + W); ++r) for (int k = 0; k < reuse; ++k) blk.push_back(r);
            std::shuffle(blk.begin(), blk.end(), rng);
            w.insert(w.end(), blk.begin(), blk.end());
        }
        char nm[64]; snprintf(nm, sizeof nm, "window-local, W = %d rows", W);
        run(nm, w);
    }
    return 0;
}
