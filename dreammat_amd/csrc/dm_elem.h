// Element type of the 16-bit activations / weights of the net kernels (conv.hip, conv_small.hip, groupnorm.hip, transformer.hip,
// attention.hip, attn_w64.hip, attn_w128.hip).  Each of those files is compiled TWICE by csrc/build.py: as written (bf16, the
// names of include/dreammat_hip.h) and with -DDM_F16 (IEEE half: what the reference's nets run in, dreammat_guidance.py:56,92-94;
// BASELINE configs[4] "fp16 UNet"), where every exported name carries f16 instead of bf16 (DM_T) or an _f16 suffix (DM_S).
// v_mfma_f32_32x32x16_f16 has the operand layout and the rate of the bf16 instruction, so tiles, LDS images, swizzles and
// schedules are shared; what differs is the 16 <-> 32 bit conversion and the exponent range a softmax numerator may use.
#pragma once

#if defined(DM_F16)
typedef _Float16 elem_t;
#define DM_T(pre, post) pre##f16##post
#define DM_S(name) name##_f16
#define DM_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define DM_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define DM_MFMA_ASM "v_mfma_f32_32x32x16_f16"
#define DM_FDOT2(a, b, c) __builtin_amdgcn_fdot2(a, b, c, false)
#define dm_attn dm_attn_f16                 // (namespace of the attention kernels' cross-file symbols)
// 2^e bounds of an un-normalised softmax numerator that is rounded to elem_t: half overflows at 65504 and loses precision below
// its smallest normal 2^-14
#define DM_P_SUM_MAX 0x1p15f                // a row sum below this proves that no numerator overflowed
#define DM_P_SUM_MIN 0x1p-8f                // a row whose sum stayed below this was scaled into the subnormals by a shared shift
#define DM_P_REBASE_AT 0x1p8f               // re-base a row once its sum passes this ...
#define DM_P_REBASE_IF 0x1p4f               // ... (rows of the pair above this are shifted along)
#else
typedef __bf16 elem_t;
#define DM_T(pre, post) pre##bf16##post
#define DM_S(name) name
#define DM_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define DM_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define DM_MFMA_ASM "v_mfma_f32_32x32x16_bf16"
#define DM_FDOT2(a, b, c) __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false)
#define DM_P_SUM_MAX 0x1p100f
#define DM_P_SUM_MIN 0x1p-100f
#define DM_P_REBASE_AT 0x1p30f
#define DM_P_REBASE_IF 0x1p20f
#endif

typedef elem_t elem8 __attribute__((ext_vector_type(8)));
typedef elem_t elem4 __attribute__((ext_vector_type(4)));
typedef elem_t elem2 __attribute__((ext_vector_type(2)));

// the two elements packed in one dword, as fp32
__device__ __forceinline__ float dm_elem_lo(unsigned w) {
#if defined(DM_F16)
    return (float)__builtin_bit_cast(elem2, w)[0];
#else
    return __builtin_bit_cast(float, w << 16);
#endif
}
__device__ __forceinline__ float dm_elem_hi(unsigned w) {
#if defined(DM_F16)
    return (float)__builtin_bit_cast(elem2, w)[1];
#else
    return __builtin_bit_cast(float, w & 0xffff0000u);
#endif
}

// ds_read_b64_tr_b16 (the transposing LDS read) moves 16-bit words whatever they encode: one builtin for both element types
__device__ __forceinline__ elem4 dm_ds_read_tr16_b64(const char* p) {
    typedef __bf16 raw4 __attribute__((ext_vector_type(4)));
    typedef raw4 __attribute__((address_space(3))) * lds4;
    return __builtin_bit_cast(elem4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4) const_cast<char*>(p)));
}
