// Shared argument block / vector types of the attention kernels (attention.hip, attn_w64.hip).
#pragma once
#include "dm_common.h"
#include "dm_elem.h"

namespace dm_attn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct AttnArgs {
    const elem_t* q; const elem_t* k; const elem_t* vt; elem_t* out;
    long long q_bs, q_ss, q_hs;       // element strides: batch, sequence, head (d contiguous)
    long long k_bs, k_ss, k_hs;
    long long vt_bs, vt_hs, vt_ds;    // V^T: batch, head, d-row (kv contiguous)
    long long o_bs, o_ss, o_hs;
    int B, Hh, Sq, Skv, D;
    float scale_log2;                 // softmax scale * log2(e)
    const elem_t* v_nat;              // training forward: V as [B, Skv, Hh, D] with K's strides (transposing LDS reads), else null -> vt
    float* lse;                       // training forward (dm_attention_fwd_lse_bf16): rowmax + log2(rowsum), log2 domain, [B, Hh, Sq]; else null
    unsigned long long* timeline;     // DREAMMAT_ATTN_TIMELINE (development): s_memtime stamps of the w64 kernel, else null
};

// attn_w64.hip: the one-wave-per-SIMD kernel (D = 64, Skv % 64 == 0)
bool attn_w64_ok(const AttnArgs& a);
int launch_attn_w64(const AttnArgs& a, hipStream_t stream);
// attn_w128.hip: the same with 128 query rows per wave (D = 64, Skv % 64 == 0, Sq % 512 == 0)
bool attn_w128_ok(const AttnArgs& a);
int launch_attn_w128(const AttnArgs& a, hipStream_t stream);

}  // namespace dm_attn
