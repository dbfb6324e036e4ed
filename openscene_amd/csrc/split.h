// Split-bf16 helpers shared by the convolution kernels: x = h1 + h2 + h3 with three bf16 pieces (exact; round to nearest).
#pragma once
#include "common.h"

namespace osn {

typedef __bf16 split_bf16x4 __attribute__((ext_vector_type(4)));

// fp32 x 4 -> the three bf16 pieces of every element (x = h1 + h2 + h3 exactly; round to nearest), two elements per instruction
__device__ __forceinline__ void tl_split4(const float4 x, split_bf16x4& p1, split_bf16x4& p2, split_bf16x4& p3) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    const f2 v[2] = {f2{x.x, x.y}, f2{x.z, x.w}};
    u2 q1, q2, q3;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const b2 h1 = __builtin_convertvector(v[j], b2);
        const uint32_t u1 = __builtin_bit_cast(uint32_t, h1);
        const f2 r1 = v[j] - f2{__builtin_bit_cast(float, u1 << 16), __builtin_bit_cast(float, u1 & 0xFFFF0000u)};
        const b2 h2 = __builtin_convertvector(r1, b2);
        const uint32_t u2_ = __builtin_bit_cast(uint32_t, h2);
        const f2 r2 = r1 - f2{__builtin_bit_cast(float, u2_ << 16), __builtin_bit_cast(float, u2_ & 0xFFFF0000u)};
        const b2 h3 = __builtin_convertvector(r2, b2);
        q1[j] = u1; q2[j] = u2_; q3[j] = __builtin_bit_cast(uint32_t, h3);
    }
    p1 = __builtin_bit_cast(split_bf16x4, q1);
    p2 = __builtin_bit_cast(split_bf16x4, q2);
    p3 = __builtin_bit_cast(split_bf16x4, q3);
}


}  // namespace osn
