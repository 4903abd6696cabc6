// qgemm_prefill.h -- what the prefill-shaped q_gemm kernels (qgemm_skinny.hip: 17 .. 128 rows, weight-stream bound; qgemm_mfma.hip:
// 256 x 256 tiles, B decoded once per call; qgemm_prefill.hip: the row pre-pass) share with their host driver.
#pragma once
#include "qgemv_common.h"

struct PrefillArgs
{
    QMatDev m;
    const f16* a;           // [M, K] in packed K order, row stride K (stage_rows_kernel's output)
    f16* c; int ldc;
    const u16* c_invperm;
    int M, c_mode;
    // qgemm_mfma.hip, many rows: the decoded weights as MFMA fragments, written once per call by wfrag_kernel (one 32 KB slot per
    // column block and K step, exactly the W stage's LDS image); null = decode inside the GEMM
    const u8* wfrag; int wfrag_steps;
};

// qgemm_prefill.hip: per (device, stream) scratch that lives until exl2_release_scratch; kind 0 = staged activation rows,
// 1 = decoded weight fragments (two buffers of one call must not alias)
int prefill_scratch(size_t bytes, void* stream, int kind, f16** out);

// qgemm_skinny.hip (17 .. 128 rows): 0 = launched, 1 = does not apply (the many-row kernel takes the call), < 0 = error (message set)
int qgemm_skinny_launch(const PrefillArgs* p, int n, bool gptq, void* stream);     // n <= 3 matrices over the same staged rows

// qgemm_mfma.hip: 0 = launched, < 0 = error (message set)
int qgemm_mfma_launch(const PrefillArgs& p, bool gptq, void* stream);
