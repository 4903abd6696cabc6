/* exl2_hip.h -- C ABI of libexl2_hip.so: the MI355X (gfx950) implementation of ExLlamaV2's quantized forward path.
 *
 * This is the drop-in boundary.  Each entry point replaces one binding of the reference's pybind module
 * `exllamav2_ext` (reference file:line cited per function; all paths under exllamav2/exllamav2_ext/).  Signatures are
 * plain C: device pointers as void*, sizes as int, the HIP stream as void* (hipStream_t; NULL = default stream).  No
 * torch types cross this boundary.  Tensors are row-major and contiguous; "half" = IEEE fp16.
 *
 * Error convention: functions return 0 (EXL2_OK) or a negative EXL2_E_* code; exl2_last_error() returns the message of
 * the calling thread's last failure (the reference throws c10::Error via TORCH_CHECK, cpp/util.h:33-38; allocation
 * failures say "HIP out of memory", which the reference's autosplit string-matches, model.py:637-639).
 * Kernels are asynchronous on `stream`; nothing synchronizes except exl2_make_q_matrix (like QMatrix's ctor,
 * cuda/q_matrix.cu:123).  Ownership: all tensors stay owned by the caller; handles store raw device pointers.
 */
#ifndef EXL2_HIP_H
#define EXL2_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EXL2_OK             0
#define EXL2_E_INVALID     -1
#define EXL2_E_OOM         -2
#define EXL2_E_HIP         -3
#define EXL2_E_UNSUPPORTED -4

const char* exl2_last_error(void);
int exl2_abi_version(void);

/* ---- q_matrix ------------------------------------------------------------------------------------------------- */

/* make_q_matrix (ext_qmatrix.cpp:21-111 -> QMatrix ctor cuda/q_matrix.cu:49-196).
 * EXL2: q_weight int32[R,N], q_perm/q_invperm uint16[K] (nullable), q_scale int32[G,N/8], q_scale_max half[G] (already
 * multiplied by prescale/256, ext.py:336), q_groups uint16[2G] (device); gptq_* NULL.
 * GPTQ: q_weight = qweight int32[K/8,N], gptq_qzeros int32[G,N/8], gptq_scales half[G,N], gptq_g_idx_host uint32[K] on
 * the HOST or NULL (act-order: q_perm / q_invperm must then be writable uint16[K] buffers, filled here).
 * q_weight is re-laid out IN PLACE (the reference shuffles in place too, q_matrix.cu:189-195). */
int exl2_make_q_matrix(void** handle, int device, int height, int width, int groups,
                       void* q_weight, void* q_perm, void* q_invperm, void* q_scale, void* q_scale_max, void* q_groups,
                       void* gptq_qzeros, void* gptq_scales, const uint32_t* gptq_g_idx_host,
                       void* bias, void* temp_dq, int max_dq_rows, void* stream);
/* free_q_matrix (ext_qmatrix.cpp:184-193) */
int exl2_free_q_matrix(void* handle);
int exl2_q_matrix_info(void* handle, int* height, int* width, int* groups, int* is_gptq, long long* weight_bytes);
/* reconstruct (ext_qmatrix.cpp:196-210): out half[K,N], original row order */
int exl2_reconstruct(void* handle, void* out, void* stream);
/* gemm_half_q_half (ext_qmatrix.cpp:213-247 -> gemm_half_q_half_cuda cuda/q_gemm.cu:201-313): c[M,N] (+)= a[M,K] W
 * (+ bias).  clear = 0 accumulates into c (residual).  r_weights: optional MoE routing weights half[M, r_weights_stride]
 * (zero weight -> row skipped; mul_r_weights -> result scaled; cuda/q_gemm_kernel.cuh:189-200,553-558). */
int exl2_gemm_half_q_half(const void* a, void* handle, void* c, int size_m, int clear,
                          const void* r_weights, int r_weights_stride, int mul_r_weights, void* stream);
/* make_group_map (ext_qmatrix.cpp:341-361), host only: returns number of uint16 written (2K) or < 0 */
int exl2_make_group_map(const uint16_t* q_groups_host, int groups, int num_qrows, uint16_t* out, int out_len);

/* ---- norms, RoPE, activation ------------------------------------------------------------------------------------ */

/* rms_norm / rms_norm_ (ext_norm.cpp:22-85 -> rms_norm_cuda cuda/rms_norm.cu:177-249) */
int exl2_rms_norm(const void* x, const void* w, void* y, float epsilon, int rows, int dim,
                  int add_residual, int input_fp32, int output_fp32, void* stream);
/* rope_ (ext_rope.cpp:21-62 -> rope_cuda cuda/rope.cu:176-218) and rope_cuda_qk (:220-273); x_k may be NULL */
int exl2_rope_qk(void* x_q, void* x_k, const void* sin, const void* cos, int batch_size,
                 int rows_per_batch_q, int rows_per_batch_k, int head_dim, int num_heads_q, int num_heads_k,
                 int past_len, const int* past_lens, int neox_style, int sincos_size, void* stream);
/* act_mul_cuda (cuda/q_mlp.cu:238-258): x = act(x) * y */
int exl2_act_mul(void* x, const void* y, int rows, int width, int act_gelu,
                 const void* r_weights, int r_weights_stride, void* stream);

/* ---- quantized KV cache --------------------------------------------------------------------------------------------- */

/* fp16_to_q_kv / q_to_fp16_kv (ext_cache.cpp:80-269 -> cuda/cache.cu:143-497, cache_q.cuh).  wbits 4 | 6 | 8. */
int exl2_fp16_to_q_kv(const void* k_in, void* k_out, void* k_scales, const void* v_in, void* v_out, void* v_scales,
                      int batch_size, int dim, int seq_stride_tokens, int offset, int width, int page_size,
                      const int* cache_seqlens, const int* block_table, int pages_per_seq, int wbits, void* stream);
int exl2_q_to_fp16_kv(const void* k_in, void* k_out, const void* k_scales, const void* v_in, void* v_out,
                      const void* v_scales, int batch_size, int dim, int seq_stride_tokens, int offset, int width,
                      int page_size, const int* cache_seqlens, const int* block_table, int pages_per_seq, int wbits,
                      void* stream);

/* FP8 cache codec: fp16_to_fp8 / fp8_to_fp16 (ext_cache.cpp:14-78 -> cuda/cache.cu:20-142).  FP8 = upper byte of the fp16
 * (E5M2 by truncation).  in/out [batch, seq, kv_heads, head_dim]; row_stride = seq * kv_heads * head_dim elements,
 * token_size = kv_heads * head_dim; tokens [offset, offset + width) of the first batch_size rows. */
int exl2_fp16_to_fp8(const void* in, void* out, int batch_size, long long row_stride, int token_size, int offset, int width,
                     void* stream);
int exl2_fp8_to_fp16(const void* in, void* out, int batch_size, long long row_stride, int token_size, int offset, int width,
                     void* stream);

/* cache_rotate (ext_cache.h / cuda/cache.cu:499-576; defragmenter generator/dynamic.py:1350-1471): cyclic move of whole
 * pages of a paged cache: temp <- page[order[0]]; page[order[i]] <- page[order[i+1]]; page[order[n-1]] <- temp.
 * order: int32[n] on the device.  No temp page is needed here (kept in registers). */
int exl2_cache_rotate(void* cache, const int* order, long long page_bytes, int n, void* stream);

/* count_match (ext_cache.cpp:285-302): host; leading positions at which two int64 token rows agree, <= min(max_a, len_b) */
int exl2_count_match(const long long* a, const long long* b, int max_a, int len_b, int* match);

/* ---- copies between the devices of a single-process tensor-parallel split (csrc/peer.hip) --------------------------------
   The reference's tp_gather / tp_broadcast (ext_tp.cpp:129-287) bounce every exchange through a pinned host buffer; here the
   device targets are written directly over xGMI (exllamav2_amd/ext_tp.py).
   exl2_memcpy_2d_async: `height` rows of `width_bytes`, row r from src + r * spitch to dst + r * dpitch, on `stream`; either
   side may be pinned host memory, memory of the current device or of a peer device.
   exl2_enable_peer_access: peer access between every ordered pair of `devices`; returns the number of ordered pairs without
   a direct path (their copies are staged by the runtime), < 0 on error. */
/* Frees the activation staging buffers the prefill / batched-decode launches keep per (device, stream) on the CURRENT device:
   those of `stream`, or of all streams (all_streams != 0).  Only when no captured graph that used them will be replayed
   again.  Returns the bytes released (retired buffers of grown slots are freed too). */
long long exl2_release_scratch(void* stream, int all_streams);
int exl2_memcpy_2d_async(void* dst, long long dpitch, const void* src, long long spitch, long long width_bytes,
                         long long height, void* stream);
int exl2_enable_peer_access(const int* devices, int n);

/* ---- load path ------------------------------------------------------------------------------- */

/* stloader_read (ext_stloader.cpp:11-157; called by stloader.py:160 for every tensor of a checkpoint): `size` bytes at
   `offset` of `filename` -> `target`.  target_device < 0: host memory (8 readers, contiguous shares, straight into the
   tensor).  target_device >= 0: device memory of that GPU through a ring of pinned slots (allocated once per device),
   one hipMemcpyAsync per 4 MiB chunk on `stream`, overlapped with the reads; returns after the last copy has completed. */
int exl2_stloader_read(const char* filename, unsigned long long offset, unsigned long long size, void* target,
                       int target_device, void* stream);
/* tensor_remap (ext_stloader.cpp:160-184), host: in place new[r][c] = old[r][index[c]], int32 [rows, cols] */
int exl2_tensor_remap(int* tensor, int rows, int cols, const int* index);
/* tensor_remap_4bit (ext_stloader.cpp:186-219), host: the same on nibbles packed 8 per int32: tensor [rows, cols / 8] */
int exl2_tensor_remap_4bit(int* tensor, int rows, int cols, const int* index);

/* ---- attention (replaces flash_attn_with_kvcache, attn.py:602-613, and _attn_torch, attn.py:869-937) ---------------- */

long long exl2_paged_attn_scratch_bytes(int rows, int head_dim, int nsplit);
int exl2_paged_attn(const void* q, const void* k_cache, const void* v_cache, void* out,
                    const int* cache_seqlens, const int* block_table,
                    int batch, int q_len, int num_heads, int num_kv_heads, int head_dim,
                    int page_size, int pages_per_seq, int len_const, int len_offset,
                    float softmax_scale, int causal, int nsplit, void* scratch, long long scratch_bytes, void* stream);
/* the same with the two keyword arguments the reference passes to flash-attn for Mistral / Gemma-type checkpoints (attn.py:590-600):
   window_left >= 0 = flash-attn's window_size[0] (a query at absolute position p sees keys [p - window_left, p]; < 0: no window; needs
   causal), softcap > 0: scores = softcap * tanh(q.k * scale / softcap) (0: off).  Replaces attn.py:905-933's softcap_ + window slicing. */
int exl2_paged_attn_ex(const void* q, const void* k_cache, const void* v_cache, void* out,
                       const int* cache_seqlens, const int* block_table,
                       int batch, int q_len, int num_heads, int num_kv_heads, int head_dim,
                       int page_size, int pages_per_seq, int len_const, int len_offset,
                       float softmax_scale, int causal, int nsplit, void* scratch, long long scratch_bytes,
                       int window_left, float softcap, void* stream);
/* the same contract for prefill-shaped steps (many query rows per sequence): MFMA flash attention, csrc/attn_prefill.hip
   (replaces _attn_torch's SDPA / matmul route, attn.py:869-937, and flash_attn_func, attn.py:960-977).  Returns 1 (nothing
   launched) for a head_dim outside {64, 128, 256}. */
int exl2_flash_prefill(const void* q, const void* k_cache, const void* v_cache, void* out,
                       const int* cache_seqlens, const int* block_table,
                       int batch, int q_len, int num_heads, int num_kv_heads, int head_dim,
                       int page_size, int pages_per_seq, int len_const, int len_offset,
                       float softmax_scale, int causal, void* stream);
/* ... with the sliding window and softcap of exl2_paged_attn_ex; returns 1 (nothing launched) also for a paged cache whose pages are
   shorter than one 64-key tile */
int exl2_flash_prefill_ex(const void* q, const void* k_cache, const void* v_cache, void* out,
                          const int* cache_seqlens, const int* block_table,
                          int batch, int q_len, int num_heads, int num_kv_heads, int head_dim,
                          int page_size, int pages_per_seq, int len_const, int len_offset,
                          float softmax_scale, int causal, int window_left, float softcap, void* stream);
/* RoPE(q, k_new) in place + append of k_new / v_new into the (paged) cache at device-side positions */
int exl2_rope_kv_append(void* q, void* k_new, const void* v_new, void* k_cache, void* v_cache,
                        const void* sin, const void* cos, int batch, int q_len, int num_heads, int num_kv_heads,
                        int head_dim, int past_len, const int* past_lens, const int* block_table,
                        int page_size, int pages_per_seq, int rope_style, int sincos_size, void* stream);

/* One-launch decode step: RoPE(q, k_new) + append(k_new, v_new) + split-KV attention + merge == the whole
   flash_attn_with_kvcache(q, k_cache, v_cache, k = k_new, v = v_new, cache_seqlens, block_table, causal = True) call of
   attn.py:602-613 preceded by rope_ (ext_bindings.cpp:123).  Returns 1 without launching when the shape is not covered
   (then use exl2_rope_kv_append + exl2_paged_attn).  counters: zeroed u32[n_counters], left zeroed. */
int exl2_attn_decode_fused(const void* q, const void* k_new, const void* v_new, void* k_cache, void* v_cache, void* out,
                           const void* sin, const void* cos, const int* cache_seqlens, const int* block_table,
                           int batch, int q_len, int num_heads, int num_kv_heads, int head_dim,
                           int page_size, int pages_per_seq, int past_const, float softmax_scale,
                           int rope_style, int sincos_size, int nsplit, void* scratch, long long scratch_bytes,
                           void* counters, int n_counters, const void* out_invperm, void* stream);
/* out_invperm (nullable, u16 [num_heads * head_dim]): feature n of a token's output row is stored at column out_invperm[n]
   -- o_proj's packed (act-order) K order, so that exl2_q_attn_forward_2_chain copies its input without a gather. */
/* The same launch leaving TWO copies of the output: `out` through out_invperm and `out_natural` in flash-attn's order -- the
   tensor flash_attn_func returns to the reference host (attn.py:960-977), which hands it to q_attn_forward_2 (attn.py:1195-1203);
   the module chain behind the operator boundary (dropin/_exl2_fast.cpp) recognises that tensor and reads the packed copy.
   out_invperm == NULL: one natural-order copy, in out_natural (or `out` when that is NULL too). */
int exl2_attn_decode_fused_dual(const void* q, const void* k_new, const void* v_new, void* k_cache, void* v_cache, void* out,
                                const void* sin, const void* cos, const int* cache_seqlens, const int* block_table,
                                int batch, int q_len, int num_heads, int num_kv_heads, int head_dim,
                                int page_size, int pages_per_seq, int past_const, float softmax_scale,
                                int rope_style, int sincos_size, int nsplit, void* scratch, long long scratch_bytes,
                                void* counters, int n_counters, const void* out_invperm, void* out_natural, void* stream);

/* Decode attention straight from the Q4 KV cache (ExLlamaV2Cache_Q4, cache.py:409-606; format cache_q.cuh:4-185): replaces
   q_to_fp16_kv of the whole live range (cache.py:472-514) + attention over the fp16 temp.  k_new / v_new (nullable, fp16
   [b, q_len, KVH, hd], k_new rotated): the step's own keys, attended in fp16 like the reference does before it quantises them;
   NULL = all keys come from the codes.  Returns 1 without launching for shapes it does not cover. */
int exl2_paged_attn_q4(const void* q, const void* k_codes, const void* k_scales, const void* v_codes, const void* v_scales,
                       const void* k_new, const void* v_new, void* out, const int* cache_seqlens, const int* block_table,
                       int batch, int q_len, int num_heads, int num_kv_heads, int head_dim,
                       int page_size, int pages_per_seq, int len_const, int len_offset,
                       float softmax_scale, int causal, int nsplit, void* scratch, long long scratch_bytes,
                       const void* out_invperm, void* stream);
/* (out_invperm: nullable, as in exl2_attn_decode_fused -- the chained decode step over a Q4 cache) */
/* The same with tickets: counters = zeroed u32[n_counters >= batch * q_len * num_heads (one per query row always suffices; fewer than the launch needs: the combine launch runs instead)], left zeroed (the
   buffer of exl2_attn_decode_fused serves).  The split partials are merged by the last split to finish, inside the launch: one
   launch instead of attention + combine. */
int exl2_paged_attn_q4_merged(const void* q, const void* k_codes, const void* k_scales, const void* v_codes, const void* v_scales,
                              const void* k_new, const void* v_new, void* out, const int* cache_seqlens, const int* block_table,
                              int batch, int q_len, int num_heads, int num_kv_heads, int head_dim,
                              int page_size, int pages_per_seq, int len_const, int len_offset,
                              float softmax_scale, int causal, int nsplit, void* scratch, long long scratch_bytes,
                              const void* out_invperm, void* counters, int n_counters, void* stream);
/* The whole decode step over a Q4 cache in ONE launch: RoPE on q / k_new on the way into the kernel (q and k_new are NOT modified),
   Q4 pack of the rotated k_new and of v_new at positions past + j (past = cache_seqlens[b], or past_const without cache_seqlens),
   attention over the codes (keys < past) and the step's own rows in fp16, split merge by ticket.  Replaces exl2_rope_kv_append +
   exl2_fp16_to_q_kv + exl2_paged_attn_q4 of a decode step (cache.py:517-556 + attn.py:602-613 over ExLlamaV2Cache_Q4).  counters:
   zeroed u32[n_counters >= batch * q_len * num_heads (one per query row always suffices; fewer than the launch needs: the combine launch runs instead)], left zeroed.  Returns 1 without launching for
   shapes it does not cover (head_dim != 128, partial rotary): use exl2_rope_quant_append_q4 + exl2_paged_attn_q4_merged. */
int exl2_attn_q4_decode_fused(const void* q, const void* k_new, const void* v_new, void* k_codes, void* k_scales, void* v_codes,
                              void* v_scales, void* out, const void* sin, const void* cos, const int* cache_seqlens,
                              const int* block_table, int batch, int q_len, int num_heads, int num_kv_heads, int head_dim,
                              int page_size, int pages_per_seq, int past_const, float softmax_scale, int rope_style, int sincos_size,
                              int nsplit, void* scratch, long long scratch_bytes, void* counters, int n_counters,
                              const void* out_invperm, void* stream);
/* RoPE(q, k_new) in place + Q4 pack of the rotated k_new and of v_new into the cache's codes / scales at device-side positions:
   ONE launch for what a decode step over a Q4 cache otherwise does with exl2_rope_kv_append (into the fp16 staging pages) +
   exl2_fp16_to_q_kv (paged, wbits 4; ext_cache.cpp:80-173 / cache.cu:143-195) -- same fp16 arithmetic, bit-identical codes and
   scales.  Conventions of exl2_rope_kv_append; codes [pages | batch, page_size, KVH, hd / 2] u8, scales [.., KVH, hd / 32] fp16.
   Returns 1 without launching for shapes it does not cover (head_dim != 128, partial rotary). */
int exl2_rope_quant_append_q4(void* q, void* k_new, const void* v_new, void* k_codes, void* k_scales, void* v_codes, void* v_scales,
                              const void* sin, const void* cos, int batch, int q_len, int num_heads, int num_kv_heads,
                              int head_dim, int past_len, const int* past_lens, const int* block_table,
                              int page_size, int pages_per_seq, int rope_style, int sincos_size, void* stream);

/* ---- fused modules --------------------------------------------------------------------------------------------------- */

/* make_q_attn (ext_qattn.cpp:24-104), q_attn_forward_1 (:115-159), q_attn_forward_2 (:161-191) */
int exl2_make_q_attn(void** handle, const void* layernorm, const void* layernorm_bias, int layernorm_is_rms,
                     int headnorm_is_rms, float norm_epsilon, void* q_q_proj, void* q_k_proj, void* q_v_proj,
                     void* q_o_proj, void* temp_state, void* temp_dq, int max_rows, int hidden_size, int num_heads,
                     int num_kv_heads, int head_dim, int max_seq_len, int has_residual, int rope_style, int sincos_size,
                     const void* q_norm, const void* k_norm, const void* post_layernorm,
                     const void* post_layernorm_bias, int residual_fp32, int use_graphs);
int exl2_free_q_attn(void* handle);
int exl2_q_attn_forward_1(void* handle, const void* x, int batch_size, int q_len, int past_len, const int* past_lens,
                          void* temp_q, void* temp_k, void* temp_v, const void* sin, const void* cos, int apply_rope,
                          void* stream);
int exl2_q_attn_forward_2(void* handle, void* x, const void* attn_output, int batch_size, int q_len, void* stream);
/* make_q_mlp (ext_qmlp.cpp:22-85), q_mlp_forward_ (:87-118) */
int exl2_make_q_mlp(void** handle, const void* layernorm, const void* layernorm_bias, int layernorm_is_rms,
                    float norm_epsilon, void* q_gate, void* q_up, void* q_down, void* temp_state, void* temp_a,
                    void* temp_b, void* temp_dq, int max_rows, int act_gelu, int has_residual,
                    const void* post_layernorm, const void* post_layernorm_bias, int residual_fp32, int use_graphs);
int exl2_free_q_mlp(void* handle);
int exl2_q_mlp_forward(void* handle, void* x, int rows, void* stream);

/* make_q_moe_mlp / q_moe_mlp_forward_ (ext_qmlp.h, ext_qmlp.cpp:245-272 -> QMoEMLP::forward_ cuda/q_mlp.cu:318-402).
   w1 / w2 / w3: arrays of num_experts q_matrix handles (gate, down, up projections).  num_experts in {4, 8, 16}.
   The reference accepts rows <= 4; this entry point takes any rows <= max_rows (16-row passes). */
int exl2_make_q_moe_mlp(void** handle, const void* layernorm, const void* layernorm_bias, int layernorm_is_rms,
                        float norm_epsilon, const void* gate, int num_experts, int num_experts_per_token,
                        void* const* w1, void* const* w2, void* const* w3, void* temp_state, void* temp_gathered_state,
                        void* temp_a, void* temp_b, void* temp_logits, void* temp_dq, int max_rows, int act_gelu);
int exl2_free_q_moe_mlp(void* handle);
int exl2_q_moe_mlp_forward(void* handle, void* x, int rows, void* stream);
/* the same inside a chained decode step (no reference counterpart: the hand-off replaces the next module's norm launch,
   rms_norm.cu:33-175): also publishes (xp_out, ss_out with *npart_out partial sums per row) for the next consumer */
int exl2_q_moe_mlp_forward_chain(void* handle, void* x, int rows, const void* next_invperm, const void* next_norm_w, void* xp_out,
                                 float* ss_out, int* npart_out, void* stream);
/* router: logits[rows, E] = x gate^T, then softmax -> top-k -> renormalise in place (cuda/q_mlp_softmax.cuh) */
int exl2_moe_route(const void* x, const void* gate, void* logits, int rows, int hidden, int num_experts, int topk, void* stream);

/* ---- chained decode (ours; replaces the COMPOSITION q_attn.cu:153-345 / q_mlp.cu:153-236 make of rms_norm + q_gemm +
   act_mul launches on the decode path; csrc/qgemv_lean.hip for <= 4 rows, csrc/qgemv_flat.hip beyond).  A producer leaves the
   residual stream in the form its consumer's prologue wants: `xp` = x TIMES THE CONSUMER'S RMSNorm WEIGHT (fp32 product,
   one rounding to fp16, saturated), in the consumer's packed (act-order) K order [rows, hidden]; `ss` = one partial sum
   of squares of x per producer workgroup [rows, npart].  The consumer multiplies xp with its matrices and scales the
   finished fp32 sums by rsqrt(sum(ss) / hidden + eps): the per-element normalisation happens once, in the producer,
   instead of once per 16-column tile of every consumer (round-3 PMC runs: it was 30 % of a consumer wave's
   instructions), and the same number of fp16 roundings as rms_norm-then-q_gemm (rms_norm.cu:33-175) is made.
   `next_norm_w` below = the norm weight of the NEXT consumer gathered through that consumer's q_perm (exl2_*_chain_info
   return it for modules, exl2_gather_f16 makes it for a head); NULL = all ones.  A module is chain-capable when its fused input
   projections share one act-order permutation (they do in every EXL2 checkpoint: the quantizer reuses one Hessian for
   q/k/v and for gate/up, conversion/quantize.py:138-139,165) and it is a plain pre-RMSNorm residual block.
   rows <= 16.  All of these return EXL2_E_INVALID "not covered" for shapes outside the kernel's reach (nothing launched). */
int exl2_q_attn_chain_info(void* handle, int* capable, const void** in_invperm, const void** o_invperm, const void** norm_w_perm);
int exl2_q_mlp_chain_info(void* handle, int* capable, const void** in_invperm, const void** norm_w_perm);
int exl2_q_matrix_perm_info(void* q_matrix, const void** perm, const void** invperm);
/* q, k, v = proj(rmsnorm(x)) from (xp, ss); no RoPE (the attention launch rotates) */
int exl2_q_attn_forward_1_chain(void* handle, const void* xp, const float* ss, int npart, int rows,
                                void* temp_q, void* temp_k, void* temp_v, void* stream);
/* the same + RoPE(q, k) in place: q_attn_forward_1's whole contract (ext_qattn.cpp:115-159) from a published hand-off */
int exl2_q_attn_forward_1_chain_rope(void* handle, const void* xp, const float* ss, int npart, int batch_size, int q_len, int past_len,
                                     const int* past_lens, void* temp_q, void* temp_k, void* temp_v, const void* sin, const void* cos,
                                     void* stream);
/* x += attn_out . Wo with attn_out already in o_proj's packed order (exl2_attn_decode_fused out_invperm); publishes
   (xp_out, ss_out) for the next consumer through next_invperm (nullable = identity); *npart_out = partials per row
   (ss_out: room for 512 floats per row) */
int exl2_q_attn_forward_2_chain(void* handle, void* x, const void* attn_out_packed, int rows, const void* next_invperm,
                                const void* next_norm_w, void* xp_out, float* ss_out, int* npart_out, void* stream);
/* x += (act(n Wg) * (n Wu)) Wd, n from (xp, ss); publishes (xp_out, ss_out) for the next consumer */
int exl2_q_mlp_forward_chain(void* handle, void* x, const void* xp, const float* ss, int npart, int rows,
                             const void* next_invperm, const void* next_norm_w, void* xp_out, float* ss_out, int* npart_out, void* stream);
/* One half of exl2_q_mlp_forward_chain for rows [row0, row0 + rows) of a step (x / xp / ss / xp_out / ss_out point at row row0):
   part 1 = gate | up (reads xp, ss; leaves act(gate) * up of those rows in the module's scratch), part 2 = down (+ residual, chain-out).
   A decode step of 5..16 sequences groups its rows per launch by what fits in LDS (K = hidden for gate | up, K = intermediate for
   down: csrc/qgemv_lean.hip ROWS form), so the two halves may be called with different row groups.  Composition replaced: QMLP::
   forward_run_ (q_mlp.cu:153-236). */
int exl2_q_mlp_forward_chain_part(void* handle, int part, int row0, void* x, const void* xp, const float* ss, int npart, int rows,
                                  const void* next_invperm, const void* next_norm_w, void* xp_out, float* ss_out, int* npart_out,
                                  void* stream);
/* c = rmsnorm(x) . W from (xp, ss); the producer of xp applied the norm weight gathered through W's q_perm (exl2_gather_f16) */
int exl2_gemm_half_q_half_chain(const void* xp, const float* ss, int npart, float eps,
                                void* q_matrix, void* c, int rows, void* stream);
/* embedding rows -> x, and published as (xp_out, ss_out with npart = 1) for the first consumer */
/* Chained decode of 5..16 rows (round 4): between exl2_chain_set_tiled(1) and (0) the chain's hand-off buffers xp (the residual
 * stream times its consumer's norm weight, in that consumer's order) hold the layout the matrix cores read their A operand in:
 * [K / 8][16 row slots][8 halfs] -- element (row, k) at ((k >> 3) * 16 + row) * 8 + (k & 7); buffers of 16 x hidden halfs whatever
 * the number of rows.  exl2_embed_rows_chain, exl2_q_attn_forward_2_chain and exl2_q_mlp_forward_chain write xp so,
 * exl2_q_attn_forward_1_chain, exl2_q_mlp_forward_chain and exl2_gemm_half_q_half_chain read it so (the lean kernel's XMEM form:
 * one coalesced kilobyte per 16 x 32 operand, no staged copy).  Process-wide switch, set around the launches of a step (or
 * their capture).  No reference counterpart (the reference has no chained decode: q_attn.cu:153-345 / q_mlp.cu:153-236 run module
 * by module). */
int exl2_chain_set_tiled(int on);

int exl2_embed_rows_chain(const void* table, const int* ids, void* x, int rows, int hidden, int vocab,
                          const void* next_invperm, const void* next_norm_w, void* xp_out, float* ss_out, void* stream);
/* The hand-off a chained producer would have left, made from rows x [rows, hidden] that are already in memory: xp_out = x *
   next_norm_w in the consumer's packed order, ss_out[row] = sum of squares of row (npart = 1).  Run by the module chain behind
   the operator boundary when q_attn_forward_1 (ext_qattn.cpp:115-159) / q_mlp_forward_ (ext_qmlp.cpp:87-118) receive a residual
   stream the previous module did not publish for them (first token, a host that touched x in between). */
int exl2_publish_rows(const void* x, int rows, int hidden, const void* next_invperm, const void* next_norm_w,
                      void* xp_out, float* ss_out, void* stream);
/* dst[i] = src[perm[i]] (u16 perm, nullable = copy), n elements */
int exl2_gather_f16(const void* src, const void* perm, void* dst, int n, void* stream);
/* Overlapped chain (csrc/chain_sync.h; no reference counterpart -- the reference serialises every launch of a decode step on
   one stream, q_attn.cu:153-345 / q_mlp.cu:153-236).  Between _begin and _end every chained launch of the calling thread
   (exl2_q_attn_forward_1_chain, exl2_attn_decode_fused, exl2_q_attn_forward_2_chain, exl2_q_mlp_forward_chain (two launches),
   exl2_gemm_half_q_half_chain) ignores its own `stream` argument: launch k goes to stream_a / stream_b alternately, waits for
   launch k-1 through block k-1 of `flags` and publishes through block k.  `flags`: n_blocks * 320 u32 of device memory,
   zero when first used (the kernels leave them zero again).  The two streams are NOT joined here: the caller orders them at
   the boundaries of a step (events), and captures them as two graphs when it wants graphs -- the branches of one forked
   HIP graph are not executed concurrently on this stack (tools/probes/fork_probe.hip).  EXPERIMENTAL. */
int exl2_chain_overlap_begin(void* flags, int n_blocks, void* stream_a, void* stream_b);
/* how many chained q_gemm launches of this process went to the round-3 kernel (csrc/qgemv_lean.hip: <= 4 rows) and how many to
   the round-2 kernel (csrc/qgemv_flat.hip: what the former declines); reset != 0 zeroes the counters.  Diagnostics for tests and
   bench.py (no reference counterpart). */
int exl2_chain_route_counts(long long* lean, long long* flat, int reset);
int exl2_chain_overlap_end(int* n_launches);
/* which variant of the prefill q_gemm (csrc/qgemm_mfma.hip, >= 129 rows) the LAST call of this process took: out4 = {rows, tile
   rows / 32 (8 = 256-row tile, 4 = 128-row tile), 1 when the weights were decoded once per call by wfrag_kernel / 0 when inside
   the GEMM, number of calls so far}.  Diagnostics for tests that force a variant (no reference counterpart; the reference's
   dispatch is q_gemm.cu:201-313). */
int exl2_prefill_route_info(int* out4);

/* ---- decode-loop utilities and graphs (replace cuda/graph.cu and the host-side embedding / argmax round trips) ------- */

int exl2_embed_rows(const void* table, const int* ids, void* out, int rows, int hidden, int vocab, void* stream);
/* greedy sampling on the device (test_inference.py:607 argmax; first maximum wins like torch.argmax).  history (nullable):
   token log [rows, hist_stride], written at hist_pos[row] + pos_inc; pos_inc != 0 also stores that sum back to hist_pos (the
   decode loop's position increment folded into this launch). */
int exl2_argmax_rows(const void* logits, int* out_ids, int rows, int vocab, int ld,
                     int* history, int* hist_pos, int hist_stride, int pos_inc, void* stream);
int exl2_add_i32(int* p, int n, int value, void* stream);
/* temperature / top-k / top-p / min-p sampling on the device, one token per row (csrc/sampling.hip).  Replaces, for these
   settings, sample_basic (exllamav2_ext/ext_sampling.cpp:93-301; binding ext_bindings.cpp:37) and the CPU stages it calls
   (exllamav2_ext/cpp/sampling.cpp: softmax_cpu :113-192, top_k_cpu :443-520, normalize_cpu :265-281, top_p_cpu :524-566,
   min_p_cpu :620-640 + keep_threshold :569-592, multinomial_cpu :872-915) together with the logits' trip to the host
   (dynamic.py:1224-1225).  logits: [rows, ld] fp16 (logits_f32 = 0) or fp32 (1), vocab <= ld; logit_filter: nullable
   bool [rows, vocab] (0 = token excluded); random in [0, 1): the point of the first row, later rows advance it with the
   reference's recurrence (:286-296).  temperature < 0.01 means greedy (:143-147).  top_k must end up in [1, 500] and below
   vocab (the reference's heap regime) -- anything else is EXL2_E_UNSUPPORTED, never an approximation.  out_tokens int32
   [rows], out_probs fp32 [rows] (probability of the token among the final candidates), workspace fp32 [rows, vocab]. */
int exl2_sample_rows(const void* logits, int logits_f32, int rows, int vocab, int ld, const void* logit_filter,
                     float temperature, int top_k, float top_p, float min_p, float random,
                     int* out_tokens, float* out_probs, float* workspace, void* stream);
/* The same sampler as the LAST launch of a decode-step graph: the step's random point is randoms[*counter % n_randoms] (device
   memory the host fills ahead of the run), the token goes to out_tokens, to history[row, hist_pos[row] + pos_inc] and the
   position is advanced -- what exl2_argmax_rows does for greedy decoding -- so a sampled step replays from one graph with no
   per-token launch argument.  *counter is not advanced here (every row reads it): append exl2_add_i32(counter, 1, 1). */
int exl2_sample_rows_step(const void* logits, int logits_f32, int rows, int vocab, int ld, const void* logit_filter,
                          float temperature, int top_k, float top_p, float min_p,
                          const float* randoms, int n_randoms, const int* counter,
                          int* out_tokens, float* out_probs, float* workspace,
                          int* history, int* hist_pos, int hist_stride, int pos_inc, void* stream);
int exl2_graph_begin_capture(void* stream);
int exl2_graph_end_capture(void* stream, void** graph_exec);
int exl2_graph_launch(void* graph_exec, void* stream);
int exl2_graph_free(void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* EXL2_HIP_H */
