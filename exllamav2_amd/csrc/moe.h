// moe.h -- host interface of the MoE front kernel (moe.hip) beyond its C-ABI entry, used by modules.hip
#pragma once
#include "hw.h"

#define MOE_MAX_SEL 8
// argument-block copies made by the front kernel of a ONE-row step: for k = 0, 1 (the gate|up and the down launch) the block of the
// y-th selected expert e (ascending e) is copied from src[k] + e * units[k] to dst[k] + y * units[k] (16-byte units); n_sel = top-k
// sum[k]: table k gets ONE block = the first selected expert's, with units [b_lo, b_hi) and [b2_lo, b2_hi) from the second selected
// expert's (qgemv_lean.h: LeanGroupPlan.pair_sum; n_sel = 2)
struct MoeCopy { const u32x4* src[2]; u32x4* dst[2]; int units[2]; int n_sel; int sum[2], b_lo[2], b_hi[2], b2_lo[2], b2_hi[2]; };

// exl2_moe_front + the copies (cp nullable); 0 = launched, 1 = shape outside it
int moe_front_launch(const void* x, const void* norm_w, const void* gate, const void* perm, void* xn, void* xg, void* logits,
                     int rows, int hidden, int num_experts, int topk, float eps, const MoeCopy* cp, void* stream);
