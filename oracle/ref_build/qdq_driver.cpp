// qdq_driver.cpp -- C entry points over the reference's decode functions (compiled from /root/reference, see build.sh).
// TEST INFRASTRUCTURE ONLY.  What it runs, for one column of 32 K-rows at `bits` bits (= `bits` consecutive words):
// the load-time shuffle (shuffle_{b}bit_{n}, called by shuffle_kernel q_matrix.cu:21-44) followed by the kernel-side
// dequant (dequant_{b}bit_{n}, q_gemm_kernel.cuh / q_matrix.cu reconstruct) -- i.e. exactly what a weight goes through in
// the reference between the checkpoint and the multiply -- returning (q - 2^(bits-1)) per row as fp16 bit patterns.
#include "cuda_shim.h"
#include "cuda/quant/qdq_2.cuh"
#include "cuda/quant/qdq_3.cuh"
#include "cuda/quant/qdq_4.cuh"
#include "cuda/quant/qdq_5.cuh"
#include "cuda/quant/qdq_6.cuh"
#include "cuda/quant/qdq_8.cuh"

static inline uint16_t bits_of(half h) { half_uint16 u(h); return u.as_uint16; }
template <int N> static void put(uint16_t* out, half2 (&dq)[N]) { for (int i = 0; i < N; i++) { out[2 * i] = bits_of(dq[i].x); out[2 * i + 1] = bits_of(dq[i].y); } }

extern "C" {

// words: `bits` uint32 (one column, 32 rows, stride 1); out: 32 fp16 bit patterns.  Returns 0, or -1 for a bad width.
int ref_decode_column32(int bits, const uint32_t* words, uint16_t* out)
{
    uint32_t q[8];
    for (int i = 0; i < bits && i < 8; i++) q[i] = words[i];
    switch (bits)
    {
        case 2: for (int u = 0; u < 2; u++) { shuffle_2bit_16(q + u, 1); half2 dq[8]; dequant_2bit_16(q[u], dq, 1); put(out + 16 * u, dq); } return 0;
        case 3: { shuffle_3bit_32(q, 1); half2 dq[16]; dequant_3bit_32(q[0], q[1], q[2], dq, 1); put(out, dq); } return 0;
        case 4: for (int u = 0; u < 4; u++) { shuffle_4bit_8(q + u, 1); half2 dq[4]; dequant_4bit_8(q[u], dq, 1); put(out + 8 * u, dq); } return 0;
        case 5: { shuffle_5bit_32(q, 1); half2 dq[16]; dequant_5bit_32(q[0], q[1], q[2], q[3], q[4], dq, 1); put(out, dq); } return 0;
        case 6: for (int u = 0; u < 2; u++) { shuffle_6bit_16(q + 3 * u, 1); half2 dq[8]; dequant_6bit_16(q[3 * u], q[3 * u + 1], q[3 * u + 2], dq, 1); put(out + 16 * u, dq); } return 0;
        case 8: for (int u = 0; u < 4; u++) { shuffle_8bit_4(q + 2 * u, 1); half2 dq[4]; dequant_8bit_8(q[2 * u], q[2 * u + 1], dq, 1); put(out + 8 * u, dq); } return 0;
    }
    return -1;
}

// qdq_util.cuh:24-30 dq_scale: 4-bit scale code (as stored) and the pre-multiplied group maximum -> fp16 scale
uint16_t ref_dq_scale(int qs, uint16_t max_scale_bits)
{
    half_uint16 m(max_scale_bits);
    return bits_of(dq_scale(qs, m.as_half));
}

// GPTQ: one word = 8 rows of one column, zero as the kernel passes it (stored nibble + 1, q_gemm_kernel_gptq.cuh /
// q_matrix.cu:261-318): shuffle_4bit_8 + dequant_4bit_8_prep_zero + dequant_4bit_8_gptq(scaled = false) -> (q - zero)
void ref_decode_gptq8(uint32_t word, uint32_t zero, uint16_t* out)
{
    uint32_t q = word;
    shuffle_4bit_8(&q, 1);
    half2 z1z16[2], y1y16[2], dq[4];
    dequant_4bit_8_prep_zero(zero, z1z16, y1y16);
    dequant_4bit_8_gptq(q, dq, z1z16, y1y16, 1, false);
    put(out, dq);
}

// the same with the scale folded in (dequant_4bit_8_prep_zero_scale + scaled = true): the GPTQ kernel's own arithmetic
void ref_decode_gptq8_scaled(uint32_t word, uint32_t zero, uint16_t scale_bits, uint16_t* out)
{
    uint32_t q = word;
    shuffle_4bit_8(&q, 1);
    half_uint16 s(scale_bits);
    half2 z1z16[2], y1y16[2], dq[4];
    dequant_4bit_8_prep_zero_scale(zero, s.as_half, z1z16, y1y16);
    dequant_4bit_8_gptq(q, dq, z1z16, y1y16, 1, true);
    put(out, dq);
}

}  // extern "C"
