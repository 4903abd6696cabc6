// qmatrix.h -- QMatrix handle (host) and its device-side view.
// Mirrors the role of the reference's QMatrix (exllamav2_ext/cuda/q_matrix.cuh:11-83): a handle over caller-owned
// tensors; the packed weights are re-laid out IN PLACE at construction (the reference shuffles in place too).
#pragma once
#include "qlayout.h"

#define MAX_GEMV_ROWS 16          // rows handled per pass by the skinny (decode) kernel
#define MAX_FUSED_MATS 4          // matrices sharing one launch (q,k,v | gate,up)

struct QMatDev
{
    const u32*   qw;              // re-laid packed weights (EXL2 q_weight / GPTQ qweight storage)
    const u32*   tail;            // padded partial super-chunks
    const QDesc* desc;
    const u16*   chunk_group;     // [K/32] group index of every 32-row chunk
    const u16*   perm;            // packed row -> input feature (nullable = identity)
    const u32*   q_scale;         // EXL2: 4-bit scale codes [G, N/8] ; GPTQ: qzeros [G, N/8]
    const f16*   scale_src;       // EXL2: q_scale_max [G] (already * prescale/256) ; GPTQ: scales [G, N]
    const f16*   bias;            // nullable
    const f16*   scale_pad;       // EXL2: private copy of q_scale_max padded to a dword multiple (LDS-DMA source)
    // decode-kernel prologue inputs prepared at make time (one contiguous LDS-DMA each instead of gathers + arithmetic):
    const u8*    pack;            // [q_perm (K u16, if any)] [chunk -> group map (K/32 u16)], 16-byte aligned sections
    u32          pack_units;      // size of pack in 16-byte units
    u32          pack_cg_off;     // byte offset of the chunk -> group map inside pack
    const f16*   sc_tab;          // [tile][G][16] fp16 scale of every (group, column), exactly reconstruct()'s values
    const f16*   zp_tab;          // GPTQ: [tile][G][16] zero points (nibble + 1) as fp16
    int n_desc;
    int K, N, G;
    int is_gptq;
    int n_runs;                   // 0: too many sections for the streaming kernel (generic kernel only)
    int main_run;                 // index of the largest full run
    QRun runs[MAX_RUNS];
};

struct QMatrix
{
    QMatDev dev;
    int device;
    int height, width, groups;    // K, N, G
    bool is_gptq;
    // owned device allocations
    u32*   tail_buf;
    QDesc* desc_buf;
    u16*   chunk_group_buf;
    f16*   scale_pad_buf;
    u8*    pack_buf;
    f16*   sc_tab_buf;
    f16*   zp_tab_buf;
    // caller-owned (kept for reconstruct / TP splitting)
    u32* q_weight; u16* q_perm; u16* q_invperm;
    f16* temp_dq; int max_dq_rows;
    // host copies used by launch heuristics
    int max_bits;
    u16* cg_host;                 // host copy of the chunk -> group map [K / 32] (launch-time work split of qgemv_lean.hip)
    long long weight_bytes;       // algorithmic bytes (packed weights + scales + perm + group map), for roofline reports
};

// prologue transforms of the activation vector while it is staged into LDS
enum { A_PLAIN = 0, A_RMSNORM = 1, A_SILU_MUL = 2, A_GELU_MUL = 3, A_SILU = 4, A_GELU = 5 };
// epilogue
enum { C_STORE = 0, C_ACCUM = 1 };

struct GemvJob
{
    QMatDev m;
    const f16* a;                 // [M, lda] input (A_PLAIN / A_RMSNORM: x ; A_SILU_MUL: gate)
    const f16* a2;                // A_SILU_MUL: up
    const f16* norm_w;            // A_RMSNORM: weight [K]
    f16* c;                       // [M, ldc]
    const f16* r_weights;         // MoE routing weights [M, r_stride] (nullable)
    const u16* c_invperm;         // nullable: column n is written to c[row, c_invperm[n]] (the consumer's packed order)
    int lda, ldc, r_stride;
    int a_mode, c_mode;
    int mul_r_weights;
    float norm_eps;
    int rows_as_prev;             // prefill passes: same input rows, transform and packed order as the previous job of the call
                                  // (q | k | v, gate | up with one act-order permutation): its staged copy is reused
    int tile0;                    // first block index (x) of this job inside a fused launch
    int a_stride;                 // LDS row stride of the staged activations, in halfs
    int rows_per_phase;           // max K rows staged per phase
    u32 lds_scale_off, lds_zp_off, lds_cg_off, lds_rmf_off, lds_desc_off;   // byte offsets in dynamic LDS
};

struct GemvArgs
{
    GemvJob job[MAX_FUSED_MATS];
    int n_jobs;
    int M;                        // rows (<= MAX_GEMV_ROWS)
};
