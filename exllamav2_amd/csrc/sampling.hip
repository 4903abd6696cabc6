// sampling.hip -- temperature / top-k / top-p / min-p sampling on the device.
//
// Replaces the reference's CPU sampler for these settings: sample_basic (exllamav2_ext/ext_sampling.cpp:93-301) over
// softmax_cpu / top_k_cpu / top_p_cpu / min_p_cpu / normalize_cpu / multinomial_cpu (exllamav2_ext/cpp/sampling.cpp), which the
// reference reaches after copying a row of fp32 logits to the host (dynamic.py:1224-1225).  Here the logits stay in HBM (fp16 as the
// head projection wrote them, or fp32), one workgroup of 1024 threads owns a row, and only the token id and its probability
// are written.
//
// Parity with the reference is TOKEN parity for the same `random`, so the candidate set, its order and the fp32 sums that are
// compared with thresholds follow the reference exactly:
//   * softmax: first maximum, expf((l - max) * (1 / T)), p = e * (1 / sum); only the order of the 32 K-term sum differs (tree);
//   * top-k is what the reference's min-heap of (p, index) pairs leaves (sampling.cpp:484-515), found without a heap: the k-th
//     largest probability theta by a 4 x 8-bit radix select over the fp32 bit patterns; every p > theta is kept; of the entries
//     EQUAL to theta the heap keeps those inside the prefix that ends at the k-th entry >= theta, minus the lowest-indexed ones
//     evicted by the larger entries that arrive later -- i.e. the LAST m of them in that prefix, m = k - #(p > theta); order =
//     descending (p, index);
//   * normalize / top-p / min-p / multinomial run on that <= 501-entry array in LDS, in place, by one thread, in the reference's
//     fp32 operation order (they are sequential sums compared with thresholds).  top_p_cpu's heap is a stack on input sorted this
//     way (every new pair is smaller than all pairs held); keep_threshold (sampling.cpp:569-592) is reproduced including its
//     return of n + 1 entries when every entry passes (it swaps the last entry with position n, where the array still holds what
//     an earlier stage left: the raw softmax probability of token k after top-k, or the first entry top-p dropped);
//   * the random point is scaled by 0.9998 and, for rows after the first, advanced by the reference's recurrence
//     (ext_sampling.cpp:273, 286-296).
// Not built (the host refuses, never approximates): top_k = 0 or > 500 (the reference's quicksort regime has no defined tie
// order), top-a, tfs, typical, mirostat, XTC, skew, smoothing, dynamic temperature, the top-token report.
#include "hw.h"
#include "errors.h"
#include <string.h>
#include <stdlib.h>

#define SAMPLE_THREADS 1024
#define SAMPLE_KMAX 500
#define SAMPLE_KPAD 512                // >= SAMPLE_KMAX + 2, a multiple of 8
#ifndef SAMPLE_KILL
#define SAMPLE_KILL 0                  // timing experiments (results are WRONG): 1 no candidate stages, 2 no ranking of the quick select's list
#endif
#define SAMPLE_LIST 1024               // quick select: entries at or above the lower bound it can hold (more: the radix passes)

struct SampleArgs
{
    const void* logits; int ld; int vocab;
    const u8* filter;                 // [rows, vocab] bool, nullable
    float temperature; int top_k; float top_p; float min_p; float random;
    int* out_tokens; float* out_probs;
    float* ws;                        // [rows, vocab] fp32: the row's probabilities (the reference's temp_probs)
    // inside a decode-step graph (exl2_sample_rows_step): the random point comes from device memory -- randoms[*counter % n_randoms],
    // filled by the host many tokens ahead -- and the token goes where the arg-max kernel puts it (history, position increment)
    const float* randoms; const int* counter; int n_randoms;
    int* history; int* hist_pos; int hist_stride, pos_inc;
    int quick;                        // REG: quick select in front of the radix passes (EXL2_SAMPLE_QUICK=0: A/B, tests)
};

// Row walks.  A row whose length and stride are multiples of 4 elements on a 16-byte-aligned base (every real vocabulary) is
// walked in quads -- 16-byte (fp32) / 8-byte (fp16) accesses, four independent loads in flight per thread: one workgroup owns
// a row, so what bounds a pass is load latency, not bandwidth -- any other row element by element.  Ascending index per thread.
DEV f32x4 load_quad(const float* p) { return *(const f32x4*)p; }
DEV f32x4 load_quad(const f16* p) { const f16x4 h = *(const f16x4*)p; return (f32x4){(float)h.x, (float)h.y, (float)h.z, (float)h.w}; }

// fn(i, value) for every element of row[0 .. V), elements dealt to the threads quad by quad (vec) or one by one
template <typename T, typename F>
DEV void for_row(const T* row, int V, bool vec, F fn)
{
    const int t = tid();
    if (vec)
    {
        const int nq = V >> 2;
        for (int q0 = t; q0 < nq; q0 += 4 * SAMPLE_THREADS)
        {
            f32x4 v[4];
            #pragma unroll
            for (int u = 0; u < 4; u++) { const int q = q0 + u * SAMPLE_THREADS; if (q < nq) v[u] = load_quad(row + 4 * (size_t)q); }
            #pragma unroll
            for (int u = 0; u < 4; u++)
            {
                const int q = q0 + u * SAMPLE_THREADS;
                if (q < nq) { fn(4 * q, v[u].x); fn(4 * q + 1, v[u].y); fn(4 * q + 2, v[u].z); fn(4 * q + 3, v[u].w); }
            }
        }
    }
    else for (int i = t; i < V; i += SAMPLE_THREADS) fn(i, (float)row[i]);
}

// fn(i, value) for the contiguous range [c0, c1) of an fp32 row, in ascending order (c0, c1 multiples of 4 when vec)
template <typename F>
DEV void for_range(const float* row, int c0, int c1, bool vec, F fn)
{
    if (vec)
    {
        for (int i0 = c0; i0 < c1; i0 += 16)
        {
            f32x4 v[4];
            #pragma unroll
            for (int u = 0; u < 4; u++) if (i0 + 4 * u < c1) v[u] = load_quad(row + i0 + 4 * u);
            #pragma unroll
            for (int u = 0; u < 4; u++)
                if (i0 + 4 * u < c1) { const int i = i0 + 4 * u; fn(i, v[u].x); fn(i + 1, v[u].y); fn(i + 2, v[u].z); fn(i + 3, v[u].w); }
        }
    }
    else for (int i = c0; i < c1; i++) fn(i, row[i]);
}

// inclusive scan over the 64 lanes of a wave (every lane must call)
DEV u32 wave_scan_incl(u32 v)
{
    #pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
        const u32 o = shfl_idx_u32(v, (lane_id() - d) & 63);
        if (lane_id() >= d) v += o;
    }
    return v;
}

// REG: the row lives in registers from its one read to the end of the radix select (8 quads per thread: vocabularies up to
// 32768 on the quad path, no logit filter) -- the maximum, the exponentials, their normalisation and the four radix passes then
// cost no memory pass at all; the probabilities are written to the workspace once (the later stages and the caller read them
// there).  Same element -> thread assignment and the same operation order as the memory walks: results identical bit for bit.
#define SAMPLE_REG_QUADS 8
template <typename T, bool REG>
KERNEL void __launch_bounds__(SAMPLE_THREADS) sample_rows_kernel(SampleArgs a)
{
    SHARED float red_v[16];
    SHARED int   red_i[16];
    SHARED u32   hist[1024];          // radix passes: 256 digits; quick select (REG): 1024 bins of key >> 20
    SHARED u32   lst_k[SAMPLE_LIST];  // quick select: keys / indices of the entries at or above the lower bound
    SHARED int   lst_i[SAMPLE_LIST];
    SHARED u32   scan_ge[16], scan_eq[16];
    SHARED u32   sel[12];             // 0: prefix key, 1: need, 2: P (prefix end), 3: c, 4: slot counter, 5: entries == theta,
                                      // 6: quick select's lower bound, 7: its list length, 8: theta found (flag)
    SHARED float raw_p[SAMPLE_KMAX];
    SHARED int   raw_i[SAMPLE_KMAX];
    // (16-byte aligned, padded to whole batches of 8: the one-thread stages read them 8 entries at a time)
    SHARED __attribute__((aligned(16))) float cand_p[SAMPLE_KPAD];
    SHARED __attribute__((aligned(16))) int   cand_i[SAMPLE_KPAD];
    SHARED __attribute__((aligned(16))) float stk_p[SAMPLE_KPAD];
    SHARED __attribute__((aligned(16))) int   stk_i[SAMPLE_KPAD];

    const int row = bid_x(), t = tid(), V = a.vocab, K = a.top_k;
    const T* lr = (const T*)a.logits + (size_t)row * a.ld;
    const u8* fr = a.filter ? a.filter + (size_t)row * V : nullptr;
    float* ws = a.ws + (size_t)row * V;

    const bool vec = ((V & 3) == 0) && ((a.ld & 3) == 0) && ((((size_t)a.logits) & 15) == 0) && ((((size_t)a.ws) & 15) == 0);

    // (REG) quad q = t + 1024 u of the row in pr[u]
    f32x4 pr[SAMPLE_REG_QUADS];
    const int nq_reg = V >> 2;
    auto for_regs = [&](auto fn) {
        #pragma unroll
        for (int u = 0; u < SAMPLE_REG_QUADS; u++)
        {
            const int q = t + u * SAMPLE_THREADS;
            if (q < nq_reg) { fn(4 * q, pr[u].x); fn(4 * q + 1, pr[u].y); fn(4 * q + 2, pr[u].z); fn(4 * q + 3, pr[u].w); }
        }
    };
    if constexpr (REG)
    {
        #pragma unroll
        for (int u = 0; u < SAMPLE_REG_QUADS; u++)
        {
            const int q = t + u * SAMPLE_THREADS;
            pr[u] = q < nq_reg ? load_quad(lr + 4 * (size_t)q) : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        }
    }

    // ---- 1. first maximum among the unfiltered logits (softmax_cpu_nonavx2: `logits[i] > maxl`, ascending i) -----------
    float bv = -1e38f; int bi = 0x7fffffff;
    if constexpr (REG) for_regs([&](int i, float v) { if (v > bv) { bv = v; bi = i; } });
    else for_row(lr, V, vec, [&](int i, float v) { if ((!fr || fr[i]) && v > bv) { bv = v; bi = i; } });
    for (int mask = 1; mask < 64; mask <<= 1)
    {
        const float ov = shfl_xor_f32(bv, mask);
        const int oi = (int)shfl_xor_u32((u32)bi, mask);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane_id() == 0) { red_v[wave_id()] = bv; red_i[wave_id()] = bi; }
    block_sync();
    for (int w = 0; w < 16; w++)
        if (red_v[w] > bv || (red_v[w] == bv && red_i[w] < bi)) { bv = red_v[w]; bi = red_i[w]; }
    const float maxl = bv; const int maxi = bi < V ? bi : 0;      // (a row with every entry filtered has no maximum)
    block_sync();

    // ---- 2. e = expf((l - max) / T), sum, p = e / sum ------------------------------------------------------------------
    const float itemp = 1.0f / a.temperature;
    float part = 0.0f;
    if constexpr (REG)
    {
        #pragma unroll
        for (int u = 0; u < SAMPLE_REG_QUADS; u++)
        {
            if (t + u * SAMPLE_THREADS < nq_reg)
            {
                #pragma unroll
                for (int c = 0; c < 4; c++) { const float x = expf((pr[u][c] - maxl) * itemp); part += x; pr[u][c] = x; }
            }
        }
    }
    else if (vec)
    {
        const int nq = V >> 2;
        for (int q = t; q < nq; q += SAMPLE_THREADS)
        {
            const f32x4 l = load_quad(lr + 4 * (size_t)q);
            f32x4 e;
            #pragma unroll
            for (int c = 0; c < 4; c++)
            {
                float x = 0.0f;
                if (!fr || fr[4 * q + c]) { x = expf((l[c] - maxl) * itemp); part += x; }
                e[c] = x;
            }
            *(f32x4*)(ws + 4 * (size_t)q) = e;
        }
    }
    else for (int i = t; i < V; i += SAMPLE_THREADS)
    {
        float e = 0.0f;
        if (!fr || fr[i]) { e = expf(((float)lr[i] - maxl) * itemp); part += e; }
        ws[i] = e;
    }
    for (int mask = 1; mask < 64; mask <<= 1) part += shfl_xor_f32(part, mask);
    if (lane_id() == 0) red_v[wave_id()] = part;
    block_sync();
    float esum = 0.0f;
    for (int w = 0; w < 16; w++) esum += red_v[w];
    const float isum = 1.0f / esum;
    u32 kmax = 0;                                                   // the thread's largest probability (bit pattern): quick select
    auto kmax_of = [&](float p) { const u32 key = f32_bits(p); kmax = key > kmax ? key : kmax; };
    if constexpr (REG)
    {
        #pragma unroll
        for (int u = 0; u < SAMPLE_REG_QUADS; u++)
        {
            const int q = t + u * SAMPLE_THREADS;
            if (q < nq_reg)
            {
                pr[u].x *= isum; pr[u].y *= isum; pr[u].z *= isum; pr[u].w *= isum;
                *(f32x4*)(ws + 4 * (size_t)q) = pr[u];
            }
        }
    }
    else if (vec)
        for (int q = t; q < (V >> 2); q += SAMPLE_THREADS)
        {
            f32x4 e = *(const f32x4*)(ws + 4 * (size_t)q);
            e.x *= isum; e.y *= isum; e.z *= isum; e.w *= isum;
            kmax_of(e.x); kmax_of(e.y); kmax_of(e.z); kmax_of(e.w);
            *(f32x4*)(ws + 4 * (size_t)q) = e;
        }
    else for (int i = t; i < V; i += SAMPLE_THREADS) { const float p = ws[i] * isum; kmax_of(p); ws[i] = p; }
    if (t == 0) { sel[0] = 0; sel[1] = (u32)K; sel[4] = 0; sel[6] = 0; sel[7] = 0; sel[8] = 0; }
    hist[t] = 0;                       // (the quick select's bins: zeroed in front of THIS barrier, one barrier less there)
    block_sync();                      // (workgroup-scope: the row's probabilities are visible to every thread from here)

    if (K == 1)
    {
        // top_k_cpu's greedy branch (sampling.cpp:459-481): the arg-max alone; normalize_cpu on one entry
        if (t == 0)
        {
            const float p = ws[maxi];
            a.out_tokens[row] = maxi;
            a.out_probs[row] = p * (1.0f / p);
        }
        return;
    }

    // ---- 3. theta = k-th largest probability ------------------------------------------------------------------------------
    // (a) quick select.  The K largest of the 1024 per-thread maxima are K distinct
    // entries of the row, so the lower edge t0 of the 1 / 8-octave bin that holds the K-th largest thread maximum satisfies
    // #(p >= t0) >= K -- and with the entries dealt to the threads quad by quad that count is seldom much more than K.  ONE
    // histogram add per thread (1024 bins of key >> 20), a descending scan, then every entry >= t0 goes to a list in LDS, where
    // theta, the tie counts and every candidate's rank in descending (p, index) order are found by comparing the list with
    // itself.  7 barriers and no pass over memory (one where the row does not live in registers), against 16 barriers + three
    // passes (seven) for (b); identical theta / m / candidates (both are exact selections).  A list that would overflow (flat
    // rows, tiny vocabularies) takes (b).
    u32 theta = 0, m = 0, eq_total = 0;
    bool have_theta = false, placed = false;
    if (a.quick)
    {
        if constexpr (REG) for_regs([&](int, float p) { kmax_of(p); });
        atomic_add_u32(&hist[(kmax >> 20) < 1023u ? (kmax >> 20) : 1023u], 1u);     // (p <= 1: bins 0 .. 1016; a NaN row must not leave the array)
        block_sync();
        {
            const u32 h = hist[1023 - t];
            const u32 inc = wave_scan_incl(h);
            if (lane_id() == 63) scan_ge[wave_id()] = inc;
            block_sync();
            u32 above = inc - h;
            for (int w = 0; w < wave_id(); w++) above += scan_ge[w];
            if (above < (u32)K && above + h >= (u32)K) sel[6] = (u32)(1023 - t) << 20;
        }
        block_sync();
        const u32 t0 = sel[6];
        auto to_list = [&](int i, float p)
        {
            const u32 key = f32_bits(p);
            if (key >= t0)
            {
                const u32 slot = atomic_add_u32(&sel[7], 1u);
                if (slot < (u32)SAMPLE_LIST) { lst_k[slot] = key; lst_i[slot] = i; }
            }
        };
        if constexpr (REG) for_regs(to_list); else for_row(ws, V, vec, to_list);      // (memory route: ONE pass over the row)
        block_sync();
        const u32 L = sel[7];
        if (L <= (u32)SAMPLE_LIST && L >= (u32)K)
        {
            u32 my_key = 0, rank = 0; int my_i = 0;
            if ((u32)t < L)
            {
                my_key = lst_k[t]; my_i = lst_i[t];
                u32 gt = 0, eq = 0, eq_hi = 0;
                if (SAMPLE_KILL & 2) { gt = (u32)t; eq = 1; }
                else for (u32 j = 0; j < L; j++)
                {
                    const u32 kj = lst_k[j];
                    gt += kj > my_key;
                    if (kj == my_key) { eq++; eq_hi += lst_i[j] > my_i; }
                }
                rank = gt + eq_hi;                                 // position in descending (p, index) order
                if (gt < (u32)K && gt + eq >= (u32)K) { sel[0] = my_key; sel[1] = (u32)K - gt; sel[5] = eq; sel[8] = 1u; }
            }
            block_sync();
            if (sel[8])
            {
                theta = sel[0]; m = sel[1]; eq_total = sel[5]; have_theta = true;
                if (eq_total == m)
                {
                    // no tie across the k-th place: the candidates are the list's entries >= theta, already ranked
                    if ((u32)t < L && my_key >= theta) { cand_p[rank] = __builtin_bit_cast(float, my_key); cand_i[rank] = my_i; }
                    placed = true;
                }
            }
        }
    }
    // (b) radix select on the bit patterns (p >= 0: integer order = float order), four 8-bit passes
    if (!have_theta)
    {
        // A thread adds a run of equal digits with ONE LDS atomic: in the first pass nearly every entry of a row has the same top
        // byte, and one atomic per entry would serialise 32 K adds on a single LDS word.
        u32 prefix = 0, maskbits = 0, need = (u32)K;
        for (int pass = 0; pass < 4; pass++)
        {
            const int shift = 24 - 8 * pass;
            if (t < 256) hist[t] = 0;
            block_sync();
            u32 run_digit = 0, run_len = 0;
            auto digit = [&](int, float p)
            {
                const u32 key = f32_bits(p);
                if ((key & maskbits) != prefix) return;
                const u32 d = (key >> shift) & 255;
                if (d != run_digit && run_len) { atomic_add_u32(&hist[run_digit], run_len); run_len = 0; }
                run_digit = d; run_len++;
            };
            if constexpr (REG) for_regs(digit); else for_row(ws, V, vec, digit);
            if (run_len) atomic_add_u32(&hist[run_digit], run_len);
            block_sync();
            // the digit whose bucket holds the need-th largest key: buckets taken from 255 down, one per thread of waves 0-3, a
            // wave scan + four wave totals (a single thread walking the 256 counters costs ~12 us per pass in LDS round trips)
            u32 h = 0, inc = 0;
            if (t < 256)
            {
                h = hist[255 - t];
                inc = wave_scan_incl(h);
                if (lane_id() == 63) scan_ge[wave_id()] = inc;
            }
            block_sync();
            if (t < 256)
            {
                u32 above = inc - h;                                   // keys in buckets with a larger digit
                for (int w = 0; w < wave_id(); w++) above += scan_ge[w];
                if (above < need && above + h >= need) { sel[0] = prefix | ((u32)(255 - t) << shift); sel[1] = need - above; sel[5] = h; }
            }
            block_sync();
            prefix = sel[0]; need = sel[1]; maskbits |= 0xFFu << shift;
        }
        theta = prefix;
        m = need;                          // entries equal to theta that survive
        eq_total = sel[5];                 // entries equal to theta in the row (the last pass' bucket = all 32 bits)
    }

    if (!placed)
    {
        // ---- 4. which of the entries equal to theta: the last m inside the prefix ending at the k-th entry >= theta ----------
        // (when ALL of them survive -- no tie across the k-th place, the usual case -- the set is simply every entry >= theta: one
        // gather, from the registers where the row lives there; the index-ordered passes below run only for a real tie)
        if (eq_total == m)
        {
            auto take = [&](int i, float p)
            {
                if (f32_bits(p) >= theta)
                {
                    const u32 slot = atomic_add_u32(&sel[4], 1u);
                    if (slot < (u32)SAMPLE_KMAX) { raw_p[slot] = p; raw_i[slot] = i; }
                }
            };
            if constexpr (REG) for_regs(take); else for_row(ws, V, vec, take);
            block_sync();
        }
        else
        {
            const int chunk = (((V + SAMPLE_THREADS - 1) / SAMPLE_THREADS) + 3) & ~3;      // contiguous share of a thread, in whole quads
            const int c0 = min(t * chunk, V), c1 = min(c0 + chunk, V);
            u32 n_ge = 0, n_eq = 0;
            for_range(ws, c0, c1, vec, [&](int, float p) { const u32 key = f32_bits(p); n_ge += key >= theta; n_eq += key == theta; });
            const u32 inc_ge = wave_scan_incl(n_ge), inc_eq = wave_scan_incl(n_eq);
            if (lane_id() == 63) { scan_ge[wave_id()] = inc_ge; scan_eq[wave_id()] = inc_eq; }
            block_sync();
            u32 ge_before = inc_ge - n_ge, eq_before = inc_eq - n_eq;
            for (int w = 0; w < wave_id(); w++) { ge_before += scan_ge[w]; eq_before += scan_eq[w]; }
            if (ge_before < (u32)K && ge_before + n_ge >= (u32)K)
            {
                u32 g = ge_before, e = eq_before; bool found = false;
                for_range(ws, c0, c1, vec, [&](int i, float p)
                {
                    const u32 key = f32_bits(p);
                    g += key >= theta; e += key == theta;
                    if (!found && g == (u32)K && key >= theta) { sel[2] = (u32)i; sel[3] = e; found = true; }
                });
            }
            block_sync();
            const int P = (int)sel[2];
            const u32 first_rank = sel[3] - m;             // 0-based rank (among the entries == theta, ascending index) of the first kept
            {
                u32 e = eq_before;
                for_range(ws, c0, c1, vec, [&](int i, float p)
                {
                    const u32 key = f32_bits(p);
                    bool take = key > theta;
                    if (key == theta) { take = (i <= P) && (e >= first_rank); e++; }
                    if (take)
                    {
                        const u32 slot = atomic_add_u32(&sel[4], 1u);
                        if (slot < (u32)SAMPLE_KMAX) { raw_p[slot] = p; raw_i[slot] = i; }
                    }
                });
            }
            block_sync();
        }

        // ---- 5. descending (p, index) order by rank (the pairs are distinct); position k = what top_k_cpu left there -------
        if (t < K)
        {
            const float p = raw_p[t]; const int ix = raw_i[t];
            int rank = 0;
            for (int j = 0; j < K; j++) { const float q = raw_p[j]; rank += (q > p) || (q == p && raw_i[j] > ix); }
            cand_p[rank] = p; cand_i[rank] = ix;
        }
    }
    if (t == SAMPLE_THREADS - 1) { cand_p[K] = ws[K]; cand_i[K] = K; cand_p[K + 1] = 0.0f; cand_i[K + 1] = 0; }
    block_sync();
    if (wave_id() != 0) return;
    // (wave 0 stays: lane 0 runs the reference's sequential sums and walks, the element-wise passes between them use all 64 lanes)
    const int lane = lane_id();
    const bool lead = lane == 0;

    // ---- 6. the reference's sequential stages on the candidate array, in place (one thread; sums in the reference's order) ----
    // One thread, so what a pass over the array costs is LDS latency: an entry per round trip (read, wait, add) was ~0.3 us per
    // candidate over the stages.  The passes read 8 entries per round trip into registers (two 16-byte reads) and keep the
    // reference's order of every sum and comparison; the stack's top two values live in registers.
    int n = K;
    if (SAMPLE_KILL & 1) { if (lead) { a.out_tokens[row] = cand_i[0]; a.out_probs[row] = cand_p[0]; } return; }
    auto load8 = [&](const float* arr, int i0, float (&v)[8]) {
        const f32x4 lo = *(const f32x4*)(arr + i0), hi = *(const f32x4*)(arr + i0 + 4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    };
    auto load8i = [&](const int* arr, int i0, int (&v)[8]) {
        const u32x4 lo = *(const u32x4*)(arr + i0), hi = *(const u32x4*)(arr + i0 + 4);
        v[0] = (int)lo.x; v[1] = (int)lo.y; v[2] = (int)lo.z; v[3] = (int)lo.w; v[4] = (int)hi.x; v[5] = (int)hi.y; v[6] = (int)hi.z; v[7] = (int)hi.w;
    };
    auto normalize = [&](int cnt)                                   // sampling.cpp:265-281 (cnt: the same in every lane)
    {
        float is = 0.0f;
        if (lead)
        {
            float s = 0.0f;
            for (int i0 = 0; i0 < cnt; i0 += 8)
            {
                float v[8]; load8(cand_p, i0, v);
                #pragma unroll
                for (int e = 0; e < 8; e++) if (i0 + e < cnt) s += v[e];
            }
            is = 1.0f / s;
        }
        is = shfl_idx_f32(is, 0);
        for (int i = lane; i < cnt; i += 64) cand_p[i] *= is;
        wave_sync();
    };
    normalize(n);
    if (n > 1 && a.top_p > 0.0f && a.top_p < 1.0f)                  // sampling.cpp:524-566 (heap == stack on this order)
    {
        int top = 0;                                                // the heap's content = stk[0 .. top), minimum on top
        if (lead)
        {
        float s = 0.0f;
        float tv = 0.0f, sv = 0.0f;                                 // stk_p[top - 1] and (while sv_ok) stk_p[top - 2]
        bool sv_ok = false;
        for (int i0 = 0; i0 < n; i0 += 8)
        {
            float v[8]; int vi[8];
            load8(cand_p, i0, v); load8i(cand_i, i0, vi);
            #pragma unroll
            for (int e = 0; e < 8; e++)
            {
                if (i0 + e >= n) break;
                const float p = v[e];
                if (p < 1e-6f) continue;
                if (s > a.top_p && p < tv) continue;                // (top >= 1 whenever s > 0)
                stk_p[top] = p; stk_i[top] = vi[e];
                sv = tv; sv_ok = top >= 1; tv = p; top++;
                s += p;
                while (s > a.top_p && top > 1)
                {
                    s -= tv; top--;
                    tv = sv_ok ? sv : stk_p[top - 1];
                    sv_ok = false;
                }
            }
        }
        }
        top = (int)shfl_idx_u32((u32)top, 0);
        wave_sync();
        // the result overwrites positions 0 .. top-1; everything behind keeps the previous stage's entries (min-p can reach one)
        for (int i = lane; i < top; i += 64) { cand_p[i] = stk_p[i]; cand_i[i] = stk_i[i]; }
        wave_sync();
        n = top;
        normalize(n);
    }
    if (n > 1 && a.min_p > 0.0f && a.min_p < 1.0f)                  // sampling.cpp:620-640 + keep_threshold :569-592
    {
        if (lead)
        {
        float topv = cand_p[0];
        for (int i = 1; i < n; i++) if (cand_p[i] > topv) topv = cand_p[i];
        const float thr = topv * a.min_p;
        int i = 0, j = n - 1;
        while (j >= i)
        {
            while (cand_p[i] >= thr && j >= i) i++;
            if (cand_p[j] >= thr)
            {
                const float tp_ = cand_p[i]; cand_p[i] = cand_p[j]; cand_p[j] = tp_;
                const int ti_ = cand_i[i]; cand_i[i] = cand_i[j]; cand_i[j] = ti_;
                i++;
            }
            j--;
        }
        n = i;
        }
        n = (int)shfl_idx_u32((u32)n, 0);
        wave_sync();
        normalize(n);
    }
    if (!lead) return;
    float random = a.randoms ? a.randoms[(unsigned)a.counter[0] % (unsigned)a.n_randoms] : a.random;
    for (int r = 0; r < row; r++)                                   // ext_sampling.cpp:286-296, once per earlier row
    {
        float x = random;
        for (int j = 0; j < 10; j++) { x = (float)((double)x + (1.337 + (double)random)); x *= x; x = fmodf(x, 1.0f); }
        random = x;
    }
    const float radj = (float)((double)random * 0.9998);            // :273
    int idx = 0;
    float accum = 0.0f;
    bool done = false;
    for (int i0 = 0; i0 < n && !done; i0 += 8)                       // sampling.cpp:894-906, 8 entries per LDS round trip
    {
        float v[8]; load8(cand_p, i0, v);
        #pragma unroll
        for (int e = 0; e < 8; e++)
        {
            if (done || i0 + e >= n) break;
            idx = i0 + e;
            accum = idx == 0 ? v[e] : accum + v[e];
            if (accum >= radj) { done = true; break; }
            if (idx == n - 1) { while (idx > 0 && cand_p[idx] == 0.0f) idx--; done = true; break; }
        }
    }
    a.out_tokens[row] = cand_i[idx];
    a.out_probs[row] = cand_p[idx];
    if (a.hist_pos)
    {
        const int pos = a.hist_pos[row] + a.pos_inc;
        if (a.history) a.history[(size_t)row * a.hist_stride + pos] = cand_i[idx];
        if (a.pos_inc) a.hist_pos[row] = pos;
    }
}

extern "C" {

static int sample_launch(SampleArgs a, int logits_f32, int rows, float random, void* stream)
{
    EXL2_REQUIRE(a.logits && a.out_tokens && a.out_probs && a.ws, "sample_rows: null argument");
    EXL2_REQUIRE(a.vocab >= 2 && a.ld >= a.vocab, "sample_rows: vocab %d, row stride %d", a.vocab, a.ld);
    if (a.temperature < 0.01f) { a.temperature = 1.0f; a.top_k = 1; }                 // ext_sampling.cpp:143-147
    if (a.top_k < 1 || a.top_k > SAMPLE_KMAX || a.top_k >= a.vocab)
        EXL2_FAIL(EXL2_E_UNSUPPORTED, "sample_rows: top_k %d outside [1, %d] (and < vocab): the reference's heap regime is what is built",
                  a.top_k, SAMPLE_KMAX);
    EXL2_REQUIRE(a.randoms || (random >= 0.0f && random < 1.0f), "sample_rows: random %f not in [0, 1)", (double)random);
    if (rows <= 0) return EXL2_OK;
    a.random = random;
    // the row in registers (see the kernel): quad path, <= 8 quads per thread, no filter
    int reg_on = 1;                                          // (read per call: the tests compare both routes)
    if (const char* e = getenv("EXL2_SAMPLE_REG")) reg_on = atoi(e);
    const bool vec = ((a.vocab & 3) == 0) && ((a.ld & 3) == 0) && ((((size_t)a.logits) & 15) == 0) && ((((size_t)a.ws) & 15) == 0);
    const bool reg = reg_on && vec && !a.filter && a.vocab <= 4 * SAMPLE_REG_QUADS * SAMPLE_THREADS;
    a.quick = 1;
    if (const char* e = getenv("EXL2_SAMPLE_QUICK")) a.quick = atoi(e);
    if (logits_f32) { if (reg) LAUNCH((sample_rows_kernel<float, true>), dim3((unsigned)rows), dim3(SAMPLE_THREADS), 0, stream, a);
                      else     LAUNCH((sample_rows_kernel<float, false>), dim3((unsigned)rows), dim3(SAMPLE_THREADS), 0, stream, a); }
    else            { if (reg) LAUNCH((sample_rows_kernel<f16, true>), dim3((unsigned)rows), dim3(SAMPLE_THREADS), 0, stream, a);
                      else     LAUNCH((sample_rows_kernel<f16, false>), dim3((unsigned)rows), dim3(SAMPLE_THREADS), 0, stream, a); }
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}

int exl2_sample_rows(const void* logits, int logits_f32, int rows, int vocab, int ld, const void* logit_filter,
                     float temperature, int top_k, float top_p, float min_p, float random,
                     int* out_tokens, float* out_probs, float* workspace, void* stream)
{
    SampleArgs a;
    memset(&a, 0, sizeof(a));
    a.logits = logits; a.ld = ld; a.vocab = vocab; a.filter = (const u8*)logit_filter;
    a.temperature = temperature; a.top_k = top_k; a.top_p = top_p; a.min_p = min_p;
    a.out_tokens = out_tokens; a.out_probs = out_probs; a.ws = workspace;
    return sample_launch(a, logits_f32, rows, random, stream);
}

// The same sampler as the last launch of a decode-step graph: the random point of the step is randoms[*counter % n_randoms]
// (device memory: the host fills randoms ahead of the run and nothing of the launch changes from token to token, so the step
// replays from ONE graph), the token is written to out_tokens, logged at history[row, hist_pos[row] + pos_inc] and the position
// is advanced -- exactly what exl2_argmax_rows does for greedy decoding.  *counter is NOT advanced here (every row reads it):
// the graph appends exl2_add_i32(counter, 1, 1).  Values in randoms must lie in [0, 1) (host contract; not checked on the device).
int exl2_sample_rows_step(const void* logits, int logits_f32, int rows, int vocab, int ld, const void* logit_filter,
                          float temperature, int top_k, float top_p, float min_p,
                          const float* randoms, int n_randoms, const int* counter,
                          int* out_tokens, float* out_probs, float* workspace,
                          int* history, int* hist_pos, int hist_stride, int pos_inc, void* stream)
{
    EXL2_REQUIRE(randoms && counter && n_randoms >= 1, "sample_rows_step: random buffer / counter missing");
    EXL2_REQUIRE(!history || hist_pos, "sample_rows_step: history needs hist_pos");
    SampleArgs a;
    memset(&a, 0, sizeof(a));
    a.logits = logits; a.ld = ld; a.vocab = vocab; a.filter = (const u8*)logit_filter;
    a.temperature = temperature; a.top_k = top_k; a.top_p = top_p; a.min_p = min_p;
    a.out_tokens = out_tokens; a.out_probs = out_probs; a.ws = workspace;
    a.randoms = randoms; a.n_randoms = n_randoms; a.counter = counter;
    a.history = history; a.hist_pos = hist_pos; a.hist_stride = hist_stride; a.pos_inc = pos_inc;
    return sample_launch(a, logits_f32, rows, 0.0f, stream);
}

}  // extern "C"
