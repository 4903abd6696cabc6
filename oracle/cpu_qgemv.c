// cpu_qgemv.c -- CPU restatement of the reference's EXL2 decode GEMV for bench.py's cpu_baseline leg, variant A of
// BASELINE.md section 3 ("dequantize-on-the-fly GEMV reading packed weights, per token").  TEST / MEASUREMENT
// INFRASTRUCTURE ONLY: the product (exllamav2_amd/) never loads it; tests/test_oracle.py checks it against oracle/exl2.py.
//
// What it restates (the reference has no CPU path: device kernel exllamav2_ext/cuda/q_gemm_kernel.cuh:140-565 via
// q_gemm.cu:201-313; decoders cuda/quant/qdq_*.cuh; scales qdq_util.cuh:24-30; group walk q_matrix.cu:130-159):
//     y[n] = sum over groups g of  scale[g][n] * sum over the group's rows k of  x[q_perm[k]] * (code[k][n] - 2^(bits_g - 1))
// on the ON-DISK tensors: q_weight int32 [R, N] -- a chunk of 32 K-rows at b bits is b consecutive word rows,
// a column's 32 codes form an LSB-first bit stream down those b words -- with the 4-bit group scales already decoded to fp32
// by the caller (oracle/exl2.py:exl2_scales: half((s + 1)^2) * max).  fp32 accumulation, weights never materialised; the
// fp16 rounding of (code - zero) * scale that reconstruct() applies per weight is not reproduced (2^-11 relative per
// weight: the baseline measures time, parity of the port is checked against float64 to 1e-5 and against
// x @ reconstruct() to the fp16 bar).
//
// Parallelism: a persistent pool of pthreads; the 32-row chunks of a matrix are dealt out in contiguous ranges (every thread
// streams whole word rows), each thread accumulates into its own y, a second phase sums the partial vectors by column range.
// AVX2 + FMA through the compiler's vectoriser over the column loop (built with -O3 -mavx2 -mfma; refuses to start on a host
// without them).
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct
{
    const uint32_t* qw;        // [R, N]
    const float* scales;       // [G, N]
    const int32_t* perm;       // [K]: packed row k multiplies x[perm[k]]
    const int32_t* gtab;       // [G, 3]: bits, first packed row, K-rows
    int G, R, N, K;
} QMat;

typedef uint32_t v8u __attribute__((vector_size(32), aligned(4)));
typedef int32_t  v8i __attribute__((vector_size(32)));
typedef float    v8f __attribute__((vector_size(32), aligned(4)));

// one 32-row chunk at B bits: acc[c] += sum_i x[i] * code(i, c), eight columns per step (GCC vector extensions -> AVX2 / FMA;
// the compiler's own vectoriser does not vectorise this loop nest across columns)
#define CHUNK_FN(B) \
static void chunk_##B(const uint32_t* restrict w, size_t ldw, const float* restrict x, float* restrict acc, int n) \
{ \
    const uint32_t mask = (1u << B) - 1u; \
    for (int c = 0; c < n; c += 8) \
    { \
        v8u W[B]; \
        for (int j = 0; j < B; j++) W[j] = *(const v8u*)(w + (size_t)j * ldw + c); \
        v8f s0 = {0, 0, 0, 0, 0, 0, 0, 0}, s1 = s0; \
        for (int i = 0; i < 32; i++) \
        { \
            const int pos = i * B, w0 = pos >> 5, sh = pos & 31; \
            v8u v = W[w0] >> sh; \
            if (sh + B > 32) v |= W[w0 + 1 < B ? w0 + 1 : w0] << ((32 - sh) & 31); \
            const v8f f = __builtin_convertvector((v8i)(v & mask), v8f); \
            if (i & 1) s1 += x[i] * f; else s0 += x[i] * f; \
        } \
        *(v8f*)(acc + c) += s0 + s1; \
    } \
}
CHUNK_FN(2) CHUNK_FN(3) CHUNK_FN(4) CHUNK_FN(5) CHUNK_FN(6) CHUNK_FN(8)

static void chunk_any(int bits, const uint32_t* w, size_t ldw, const float* x, float* acc, int n)
{
    switch (bits)
    {
        case 2: chunk_2(w, ldw, x, acc, n); break;
        case 3: chunk_3(w, ldw, x, acc, n); break;
        case 4: chunk_4(w, ldw, x, acc, n); break;
        case 5: chunk_5(w, ldw, x, acc, n); break;
        case 6: chunk_6(w, ldw, x, acc, n); break;
        case 8: chunk_8(w, ldw, x, acc, n); break;
        default: break;
    }
}

// ---- pool ---------------------------------------------------------------------------------------------------------------
static int g_threads = 0;
static inline int g_threads_for_bar(void) { return g_threads; }
static pthread_t* g_tid = NULL;
// sense-reversing barrier: spin (a call is tens of microseconds of work per thread; a futex barrier over 128 threads costs
// more than that), yield after a while so that an over-subscribed host still makes progress
static volatile int g_bar_count = 0, g_bar_sense = 0;
static void bar_wait(int* local_sense)
{
    const int s = *local_sense = !*local_sense;
    if (__atomic_add_fetch(&g_bar_count, 1, __ATOMIC_ACQ_REL) == g_threads_for_bar())
    {
        __atomic_store_n(&g_bar_count, 0, __ATOMIC_RELAXED);
        __atomic_store_n(&g_bar_sense, s, __ATOMIC_RELEASE);
        return;
    }
    for (unsigned spins = 0; __atomic_load_n(&g_bar_sense, __ATOMIC_ACQUIRE) != s; spins++)
    {
        if (spins < 2000) __builtin_ia32_pause(); else sched_yield();
    }
}
static volatile int g_quit = 0;
static const QMat* g_m = NULL;
static const float* g_x = NULL;
static float* g_y = NULL;
static float* g_part = NULL;       // [threads][n_cap] partial outputs
static float* g_acc = NULL;        // [threads][n_cap] per-group integer-side sums
static int g_ncap = 0;
static int g_main_sense = 0;

static void run_share(int t)
{
    const QMat* m = g_m;
    const int N = m->N, T = g_threads;
    float* part = g_part + (size_t)t * g_ncap;
    float* acc = g_acc + (size_t)t * g_ncap;
    memset(part, 0, (size_t)N * sizeof(float));
    const int chunks = m->K / 32;
    const int c0 = (int)((long long)chunks * t / T), c1 = (int)((long long)chunks * (t + 1) / T);
    // walk the groups; chunk index space is cumulative over groups
    int cbase = 0, kbase = 0;
    for (int g = 0; g < m->G && cbase < c1; g++)
    {
        const int bits = m->gtab[3 * g], q0 = m->gtab[3 * g + 1], rows = m->gtab[3 * g + 2];
        const int gch = rows / 32;
        const int lo = c0 > cbase ? c0 : cbase, hi = c1 < cbase + gch ? c1 : cbase + gch;
        if (hi > lo)
        {
            memset(acc, 0, (size_t)N * sizeof(float));
            float sx = 0.0f;
            for (int c = lo; c < hi; c++)
            {
                float xs[32];
                const int k0 = kbase + (c - cbase) * 32;
                for (int i = 0; i < 32; i++) { xs[i] = g_x[m->perm[k0 + i]]; sx += xs[i]; }
                chunk_any(bits, m->qw + (size_t)(q0 + (c - cbase) * bits) * N, (size_t)N, xs, acc, N);
            }
            const float zs = (float)(1 << (bits - 1)) * sx;
            const float* sc = m->scales + (size_t)g * N;
            for (int n = 0; n < N; n++) part[n] += sc[n] * (acc[n] - zs);
        }
        cbase += gch; kbase += rows;
    }
}

static void reduce_share(int t)
{
    const int N = g_m->N, T = g_threads;
    const int n0 = (int)((long long)N * t / T), n1 = (int)((long long)N * (t + 1) / T);
    for (int n = n0; n < n1; n++)
    {
        float s = 0.0f;
        for (int u = 0; u < T; u++) s += g_part[(size_t)u * g_ncap + n];
        g_y[n] = s;
    }
}

static void* worker(void* arg)
{
    const int t = (int)(intptr_t)arg;
    int sense = 0;
    for (;;)
    {
        bar_wait(&sense);                          // a call starts
        if (g_quit) return NULL;
        run_share(t);
        bar_wait(&sense);                          // partial outputs complete
        reduce_share(t);
        bar_wait(&sense);                          // y complete
    }
}

int cpu_qgemv_init(int threads, int n_cap)
{
    if (!__builtin_cpu_supports("avx2") || !__builtin_cpu_supports("fma")) return -1;
    if (g_threads) return g_threads;
    if (threads < 1) threads = 1;
    g_ncap = (n_cap + 15) & ~15;
    g_part = (float*)aligned_alloc(64, (size_t)threads * g_ncap * sizeof(float));
    g_acc = (float*)aligned_alloc(64, (size_t)threads * g_ncap * sizeof(float));
    g_tid = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    if (!g_part || !g_acc || !g_tid) return -2;
    g_bar_count = 0; g_bar_sense = 0; g_main_sense = 0;
    g_quit = 0; g_threads = threads;
    for (int t = 1; t < threads; t++)
        if (pthread_create(&g_tid[t], NULL, worker, (void*)(intptr_t)t) != 0) return -3;
    return threads;
}

// y[N] = x[perm] . W  (fp32).  The calling thread is worker 0.
int cpu_qgemv(const QMat* m, const float* x, float* y)
{
    if (!g_threads || m->N > g_ncap || (m->K & 31)) return -1;
    g_m = m; g_x = x; g_y = y;
    bar_wait(&g_main_sense);
    run_share(0);
    bar_wait(&g_main_sense);
    reduce_share(0);
    bar_wait(&g_main_sense);
    return 0;
}

void cpu_qgemv_shutdown(void)
{
    if (!g_threads) return;
    g_quit = 1;
    bar_wait(&g_main_sense);
    for (int t = 1; t < g_threads; t++) pthread_join(g_tid[t], NULL);
    free(g_part); free(g_acc); free(g_tid);
    g_part = g_acc = NULL; g_tid = NULL; g_threads = 0;
}
