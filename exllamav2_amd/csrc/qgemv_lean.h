// qgemv_lean.h -- host interface of the round-3 chained decode q_gemm (qgemv_lean.hip); same input record as qgemv_flat.h
#pragma once
#include "qgemv_flat.h"

#define LEAN_MAX_M 4              // rows of the wave-private form: a wave stages M rows of its K slice in its own LDS area
#define LEAN_MAX_ROWS 16          // rows of the ROWS form (round 4): the workgroup stages the whole M x K rows once, when they fit in LDS

// 0: launched; 1: shape not covered (the caller falls back to qgemv_flat_launch); *wgs_out = grid size = partial sums of
// squares a chain-out launch writes per row
int qgemv_lean_launch(const FlatIn& in, void* stream, int* wgs_out);

// ---- grouped-expert launches (sparse MoE at ONE row; q_mlp.cu:316-436 is the reference's fused form for <= 4 rows) -------------
// `n_groups` identically shaped launches (one per expert), planned once: every expert's argument block lies in device memory
// (table_src); a step's MoE front kernel copies the blocks of the n_sel experts it selected to table_sel, and ONE launch with
// blockIdx.y = 0 .. n_sel - 1 reads its block from there (the graph's launch is fixed, the experts are not).
struct LeanGroupPlan
{
    void* table_src; void* table_sel;           // device: [n_groups] / [n_sel] argument blocks of block_bytes each
    int n_groups, n_sel, block_bytes;
    int S, nslots, pair, walk, grid_x; unsigned lds;
    int all_groups;                             // 2-4 rows: one launch over ALL experts' planned blocks (r_stride != 0), no copies
    int pair_sum, b_lo, b_hi, b2_lo, b2_hi;     // FlatIn.pair_sum plans: 16-byte units of a step's block that come from the SECOND selected expert
};
// what a pair_sum launch takes at launch time: the residual rows the weighted sum is added to (+ the hand-off for the next consumer, nullable)
struct LeanGroupDyn { void* c; int ldc; void* xp_out; const void* xp_invperm; const void* xp_w; float* ss_out; int ldxp; };
// out_scale[g] (nullable array): device pointer to the fp16 weight the group's finished sums are multiplied by.  0: planned;
// 1: a shape the lean kernel declines or experts whose plans differ in geometry (the caller keeps its other route)
// r_stride != 0 (2-4 rows): out_scale[g][row * r_stride] = the group's routing weight of that row; the launch then covers all groups
// and a group without a routed row leaves at entry.  mul: the finished sums are multiplied by the weight (down projections)
int qgemv_lean_group_plan(const FlatIn* ins, int n_groups, const f16* const* out_scale, int n_sel, LeanGroupPlan* plan, int r_stride = 0, int mul = 1);
int qgemv_lean_group_launch(const LeanGroupPlan* plan, void* stream, const LeanGroupDyn* dyn = nullptr);
void qgemv_lean_group_free(LeanGroupPlan* plan);
