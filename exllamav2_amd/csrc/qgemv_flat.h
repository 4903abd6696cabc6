// qgemv_flat.h -- host interface of the chained decode q_gemm (qgemv_flat.hip), used by modules.hip
#pragma once
#include "qmatrix.h"

#define FLAT_MAX_MATS 4
// A_DIRECT: `a` rows are the activations, in the matrices' packed K order.
// A_NORM_PRE: `a` rows are x * w -- the residual stream times the consumer's RMSNorm weight, rounded to fp16 by the PRODUCER
// (xp_w below), in packed order -- and `ss` holds the partial sums of squares of x: the product is scaled by
// rsqrt(mean(x^2) + eps) at the very end (fp32), so a consumer's prologue is a plain copy and the per-element
// normalisation is done once (by the producer) instead of once per 16-column tile.
enum { A_DIRECT = 0, A_NORM_PRE = 1 };

struct FlatIn
{
    const QMatrix* qm[FLAT_MAX_MATS]; f16* c[FLAT_MAX_MATS]; const u16* c_invperm[FLAT_MAX_MATS]; int ldc[FLAT_MAX_MATS];
    int n_mats, M;
    int a_mode;                   // A_DIRECT / A_NORM_PRE
    const f16* a; int lda;
    const float* ss; int npart;   // A_NORM_PRE: partial sums of squares [M, npart]
    float eps;
    int pair;                     // 2 matrices (gate, up): output = act(gate) * up, written through mat 0's c / c_invperm
    int act_gelu, c_mode;         // C_STORE / C_ACCUM (residual)
    f16* xp_out; const u16* xp_invperm; float* ss_out; int ldxp;     // chain-out (nullable): x in the next consumer's order ...
    const f16* xp_w;              // ... times that consumer's norm weight (in ITS packed order; nullable = 1), + partial sums of x^2
    // overlapped chain (chain_sync.h / hw.h): sync_wait = the producer launch's block (its "go" word is polled before the
    // activations are read, with agent-scope loads), sync_signal = this launch's block (outputs are agent-scope stores;
    // every combining wave arrives there, the last publishes "go").  Both nullable.
    const u32* sync_wait; u32* sync_signal;
    u32* sync_arrive;             // first launch of a chain: every workgroup adds 1 on entry (chain_sync.h: the gate)
    // 5 .. 16 rows between two launches of one module (gate | up -> down, round 4): the activations in the layout the matrix
    // cores read them in -- [K / 8][16 rows][8 halfs], element (row, k) at ((k >> 3) * 16 + row) * 8 + (k & 7): the 16 x 32
    // A operand of one MFMA is 1 KB of consecutive memory, one fully coalesced request per wave (row-major: 16 separate
    // 64-byte pieces).  c_tiled: this launch WRITES its (one) output so; a_tiled: it READS `a` so (lean kernel, XMEM form only).
    int a_tiled, c_tiled;
    int xp_tiled;                 // chain-out: xp_out is written in that layout too (16 row slots; ldxp unused)
    int plan_only;                // qgemv_lean_launch: make the host plan, launch nothing (0: the lean kernel takes this shape, 1: it declines)
    int pair_sum;                 // pair = tile u of TWO experts' down projections, output = weighted sum into the residual (lean MoE route; lean_export only)
    void* lean_export;            // qgemv_lean_launch: hand the finished argument block + geometry to this LeanExport (qgemv_lean.hip), launch nothing
};

// 0: launched; 1: shape not covered; < 0: error.  *wgs_out = grid size = partial sums a chain-out launch writes per row
int qgemv_flat_launch(const FlatIn& in, void* stream, int* wgs_out);

// `n_groups` sets of identically shaped matrices (MoE experts) on blockIdx.y of one launch: group g reads its rows at
// a + g * a_gstride (A_DIRECT only), writes through its own c pointers, is skipped when no row carries a weight in column g
// of r_weights [M, r_stride]; rows with zero weight are not written; mul_r: result *= weight.  0: launched, 1: not covered.
int qgemv_flat_group_launch(const FlatIn* ins, int n_groups, const f16* r_weights, int r_stride, int mul_r, long long a_gstride,
                            void* stream);
