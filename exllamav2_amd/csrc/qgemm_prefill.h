// qgemm_prefill.h -- what the two prefill-shaped q_gemm kernels (qgemm_prefill.hip: 128 x 128 tiles, B decoded in registers;
// qgemm_mfma.hip: 256 x 256 tiles, B decoded once per workgroup into LDS) share with their host driver.
#pragma once
#include "qgemv_common.h"

struct PrefillArgs
{
    QMatDev m;
    const f16* a;           // [M, K] in packed K order, row stride K (stage_rows_kernel's output)
    f16* c; int ldc;
    const u16* c_invperm;
    int M, c_mode;
};

// qgemm_mfma.hip: 0 = launched, < 0 = error (message set)
int qgemm_mfma_launch(const PrefillArgs& p, bool gptq, void* stream);
