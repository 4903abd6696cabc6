// qgemv_lean.h -- host interface of the round-3 chained decode q_gemm (qgemv_lean.hip); same input record as qgemv_flat.h
#pragma once
#include "qgemv_flat.h"

#define LEAN_MAX_M 4              // rows of the wave-private form: a wave stages M rows of its K slice in its own LDS area
#define LEAN_MAX_ROWS 16          // rows of the ROWS form (round 4): the workgroup stages the whole M x K rows once, when they fit in LDS

// 0: launched; 1: shape not covered (the caller falls back to qgemv_flat_launch); *wgs_out = grid size = partial sums of
// squares a chain-out launch writes per row
int qgemv_lean_launch(const FlatIn& in, void* stream, int* wgs_out);
