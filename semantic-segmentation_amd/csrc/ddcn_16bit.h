// Pieces shared by the 16-bit translation units (ddcn_bf16.hip, ddcn_fused16.hip): storage traits of the two 16-bit
// formats, the bilinear blend of packed pairs, LDS-DMA helpers of the MFMA GEMMs.  gfx950 only.
#pragma once
#include "ddcn_common.h"

namespace ddcn {
namespace {

typedef unsigned short bf16_t;  // storage type at the boundary
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// The 16-bit format of a call: storage is always unsigned short at the boundary.  one(): element -> float;
// lo()/hi(): the two elements of a packed register; T::pack2(): two floats -> packed pair, round to nearest even
// (v_cvt_pk_bf16_f32 / v_cvt_f16_f32 + pack on gfx950); mfma() / mfma16(): one 32x32x16 / 16x16x32 step with fp32
// accumulation (register-staged GEMM / LDS-DMA GEMMs).
struct BF16 {
    static __device__ __forceinline__ float one(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
    static __device__ __forceinline__ float lo(unsigned u) { return __uint_as_float(u << 16); }
    static __device__ __forceinline__ float hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
    static __device__ __forceinline__ unsigned pack2(float a, float b)
    {
        const f32x2 v = {a, b};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    }
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c,
                                                       0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
struct F16 {
    static __device__ __forceinline__ float one(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
    static __device__ __forceinline__ float lo(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[0]; }
    static __device__ __forceinline__ float hi(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[1]; }
    static __device__ __forceinline__ unsigned pack2(float a, float b)
    {
        const f32x2 v = {a, b};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
    }
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c,
                                                      0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// w1*a1 + w2*a2 + w3*a3 + w4*a4 for the two bf16 halves of a register pair, rounded back to a bf16 pair
template <typename T>
__device__ __forceinline__ unsigned blend_pair(unsigned a1, unsigned a2, unsigned a3, unsigned a4, float w1, float w2,
                                               float w3, float w4)
{
    const float lo = w1 * T::lo(a1) + w2 * T::lo(a2) + w3 * T::lo(a3) + w4 * T::lo(a4);
    const float hi = w1 * T::hi(a1) + w2 * T::hi(a2) + w3 * T::hi(a3) + w4 * T::hi(a4);
    return T::pack2(lo, hi);
}

typedef __attribute__((address_space(3))) void lds_void;
constexpr int DK = 32;
constexpr unsigned DMA_OOB = 0x80000000u;

// A descriptor the compiler knows to be wave-uniform (otherwise every DMA request is wrapped in a waterfall loop)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void *p, unsigned bytes)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// LDS slot of k chunk c in a 64-byte A row: c ^ a_swizzle(row).  The table (0, 2, 3, 1) over the row quad makes the
// 16x16x32 fragment read (lane l: row l % 16, chunk l / 16) conflict-free in every 16-lane group of ds_read_b128.
__device__ __forceinline__ int a_swizzle(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

constexpr unsigned wait_vm_lgkm0(int vm) { return (unsigned)((vm & 0xF) | ((vm >> 4) << 14) | 0x0070); }


}  // namespace
}  // namespace ddcn
