// sampling_driver.cpp -- runs the REFERENCE's CPU sampler (exllamav2_ext/cpp/sampling.cpp + sampling_avx2.cpp + profiling.cpp,
// compiled from where they lie under /root/reference) in the order sample_basic calls its stages
// (exllamav2_ext/ext_sampling.cpp:137-290), for the settings the device sampler covers: temperature, top-k (heap regime,
// 1 <= k <= 500), top-p, min-p, logit filter, the batch random recurrence.  TEST INFRASTRUCTURE ONLY: pins
// oracle/sampling.py by execution; tests/golden/make_golden_sampling.py records its outputs as a fixture.
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include "cpp/sampling.h"

// defined in cpp/sampling.cpp, not declared in the header: the portable softmax (softmax_cpu picks the AVX2 twin at run time)
int softmax_cpu_nonavx2(const int vocab_size, const float temperature, const float* logits, const bool* logits_filter,
                        const float exponent, float* output);

extern "C" {

// logits fp32 [bsz, vocab]; filter (nullable) bool [bsz, vocab]; out_tokens int32 [bsz], out_probs fp32 [bsz];
// dispatch = 1: softmax_cpu (what the reference runs on this host), 0: the portable variant (bit-level pin of the oracle)
int ref_sample_basic(const float* logits, int bsz, int vocab, const uint8_t* filter, float temperature, int top_k,
                     float top_p, float min_p, float random, int dispatch, int* out_tokens, float* out_probs,
                     int* out_num_candidates)
{
    // softmax_cpu_avx2 works on the vocabulary rounded up to 32 entries (sampling_avx2.cpp:30): leave that slack
    const size_t padded = ((size_t)vocab + 63) / 32 * 32;
    float* temp_probs = (float*)aligned_alloc(32, padded * 4);
    int* temp_indices = (int*)aligned_alloc(32, padded * 4);
    if (temperature < 0.01) { temperature = 1.0f; top_k = 1; }                      // ext_sampling.cpp:143-147
    for (int i = 0; i < bsz; i++)
    {
        const bool* f = filter ? (const bool*)filter + (size_t)i * vocab : NULL;
        int maxlogit = dispatch ? softmax_cpu(vocab, temperature, logits + (size_t)i * vocab, f, 1.0f, temp_probs)
                                : softmax_cpu_nonavx2(vocab, temperature, logits + (size_t)i * vocab, f, 1.0f, temp_probs);
        for (int j = 0; j < vocab; j++) temp_indices[j] = j;
        int n = vocab;
        if (top_k > 0 && top_k < vocab) { n = top_k_cpu(n, temp_probs, temp_indices, top_k, maxlogit); normalize_cpu(n, temp_probs); }
        if (n > 1 && top_p > 0.0f && top_p < 1.0f) { n = top_p_cpu(n, temp_probs, temp_indices, top_p); normalize_cpu(n, temp_probs); }
        if (n > 1 && min_p > 0.0f && min_p < 1.0f) { n = min_p_cpu(n, temp_probs, temp_indices, min_p); normalize_cpu(n, temp_probs); }
        float random_s_adj = random * 0.9998;                                       // :273
        multinomial_cpu(n, temp_probs, temp_indices, random_s_adj);
        out_tokens[i] = temp_indices[0];
        out_probs[i] = temp_probs[0];
        if (out_num_candidates) out_num_candidates[i] = n;
        if (bsz > 1)                                                                // :286-296
        {
            float r = random;
            for (int j = 0; j < 10; ++j) { r += 1.337 + random; r *= r; r = fmod(r, 1.0f); }
            random = r;
        }
    }
    free(temp_probs); free(temp_indices);
    return 0;
}

// cpp/sampling.cpp:20-110 as ext_sampling.cpp:32-72 calls it: sequence int64 [bsz, seq_len], logits fp32 [bsz, vocab] in place
int ref_apply_rep_penalty(const uint64_t* sequence, int bsz, int seq_len, float penalty_max, int sustain, int decay,
                          float alpha_frequency, float alpha_presence, float* logits, int vocab)
{
    for (int i = 0; i < bsz; i++)
        apply_rep_penalty_cpu(vocab, sequence + (size_t)i * seq_len, penalty_max, sustain, decay, alpha_frequency, alpha_presence,
                              seq_len, logits + (size_t)i * vocab);
    return 0;
}

}  // extern "C"
