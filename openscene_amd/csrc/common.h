// Shared host/device helpers of libopenscene_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/openscene_amd.h"

namespace osn {

void set_error(const char* fmt, ...);

#define OSN_REQUIRE(cond, code, ...)            \
    do {                                        \
        if (!(cond)) {                          \
            osn::set_error(__VA_ARGS__);        \
            return (code);                      \
        }                                       \
    } while (0)

#define OSN_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess) {                                                        \
            osn::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__),      \
                           __FILE__, __LINE__);                                         \
            return OSN_E_HIP;                                                           \
        }                                                                               \
    } while (0)

#define OSN_LAUNCH_CHECK() OSN_HIP(hipGetLastError())

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- coordinate key -----------------------------------------------------------
constexpr uint64_t KEY_EMPTY = 0xFFFFFFFFFFFFFFFFull;
constexpr int COORD_BIAS = 1 << 15;

__host__ __device__ inline uint64_t pack_key(int b, int x, int y, int z) {
    return (uint64_t(uint32_t(b) & 0xFFFFu) << 48) | (uint64_t(uint32_t(x + COORD_BIAS) & 0xFFFFu) << 32) |
           (uint64_t(uint32_t(y + COORD_BIAS) & 0xFFFFu) << 16) | uint64_t(uint32_t(z + COORD_BIAS) & 0xFFFFu);
}

__host__ __device__ inline uint32_t hash_key(uint64_t k) {
    // splitmix64 finaliser
    k ^= k >> 30; k *= 0xBF58476D1CE4E5B9ull;
    k ^= k >> 27; k *= 0x94D049BB133111EBull;
    k ^= k >> 31;
    return uint32_t(k);
}

__device__ inline int floor_div(int a, int s) {
    int q = a / s;
    return (a % s != 0 && ((a < 0) != (s < 0))) ? q - 1 : q;
}

// Probe an open-addressing table; returns the stored value or -1.
__device__ inline int table_find(const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals,
                                 uint32_t mask, uint64_t key) {
    uint32_t slot = hash_key(key) & mask;
    while (true) {
        uint64_t k = keys[slot];
        if (k == key) return vals[slot];
        if (k == KEY_EMPTY) return -1;
        slot = (slot + 1) & mask;
    }
}

}  // namespace osn
