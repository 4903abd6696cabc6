// qmatrix.hip -- QMatrix construction (format validation, in-place re-layout), destruction and reconstruct.
//
// Reference: QMatrix ctor exllamav2_ext/cuda/q_matrix.cu:49-196 (reads q_groups to the host, derives the bit-width
// sections, re-shuffles q_weight in place; GPTQ: infers the group size :101-105 and builds the act-order permutation
// on the CPU :597-680), reconstruct :499-553.
#include "qmatrix.h"
#include "errors.h"
#include <vector>
#include <stdlib.h>
#include <string.h>

// ---- re-layout ------------------------------------------------------------------------------------------------------

struct RelayoutArgs
{
    const u32* src;          // copy of the on-disk tensor
    u32* dst;                // weight buffer or tail buffer
    const u16* row_map;      // GPTQ act-order: packed row -> source row (nullable)
    u32 base_word, tile_stride;
    int n_super, nvalid_last, bits, is_gptq;
    int N;
    int k_base;              // first packed K row of the run
    int src_qrow0;           // EXL2: first word-row of the section in src
    int chunk_in_sec0;       // EXL2: index of the run's first 32-row chunk within its section
    int in_tail;
};

DEV u32 stream_bits(const u32* col_words, int stride, int bitpos, int bits)
{
    const int w0 = bitpos >> 5, sh = bitpos & 31;
    u32 v = col_words[(size_t)w0 * stride] >> sh;
    if (sh + bits > 32) v |= col_words[(size_t)(w0 + 1) * stride] << (32 - sh);
    return v & ((1u << bits) - 1u);
}

KERNEL void relayout_kernel(const RelayoutArgs a)
{
    const int tile = bid_x();
    const int s = bid_y() * (nthreads() >> 6) + wave_id();
    if (s >= a.n_super) return;
    const int lane = lane_id();
    const int c = lane & 15, j = lane >> 4;
    const int n = tile * TILE_N + c;
    const int nvalid = (s == a.n_super - 1) ? a.nvalid_last : 4;

    u32 w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = 0; q < 4; q++)
    {
        for (int e = 0; e < 8; e++)
        {
            u32 code = 0;
            if (q < nvalid)
            {
                if (a.is_gptq)
                {
                    const int k = a.k_base + s * SUPER_ROWS + 32 * q + 8 * j + e;
                    const int r = a.row_map ? (int)a.row_map[k] : k;
                    code = (a.src[(size_t)(r >> 3) * a.N + n] >> (4 * (r & 7))) & 15u;
                }
                else
                {
                    const int chunk = a.chunk_in_sec0 + 4 * s + q;
                    const u32* col = a.src + (size_t)(a.src_qrow0 + chunk * a.bits) * a.N + n;
                    code = stream_bits(col, a.N, (8 * j + e) * a.bits, a.bits);
                }
            }
            put_code(w, a.bits, 4 * q + (e >> 1), e & 1, code);
        }
    }
    u32* dst = a.dst + a.base_word + (size_t)tile * a.tile_stride + (size_t)s * (64 * a.bits);
    for (int i = 0; i < a.bits; i++) dst[lane_word_index(a.bits, lane, i)] = w[i];
}

// ---- reconstruct ----------------------------------------------------------------------------------------------------

template <int BITS, bool GPTQ>
DEV void reconstruct_super(const QMatDev& m, const QDesc* dp, int tile, int s, int lane, f16* out)
{
    const u32* base = (dp->in_tail ? m.tail : m.qw) + dp->base_word + (size_t)tile * dp->tile_stride;
    LaneWords<BITS> lw;
    load_lane_words<BITS>(base + (size_t)s * (64 * BITS), lane, lw);
    const int c = lane & 15, j = lane >> 4;
    const int n = tile * TILE_N + c;
    const int nvalid = (s == dp->n_super - 1) ? dp->nvalid_last : 4;
    const int chunk0 = (dp->k_base >> 5) + 4 * s;

    f16 sc[4];
    ZC zc[4];
    #pragma unroll
    for (int q = 0; q < 4; q++)
    {
        const int g = m.chunk_group[q < nvalid ? chunk0 + q : chunk0];
        const u32 word = m.q_scale[(size_t)g * (m.N >> 3) + (n >> 3)];
        const int nib = (word >> (4 * (n & 7))) & 15;
        if constexpr (GPTQ)
        {
            sc[q] = m.scale_src[(size_t)g * m.N + n];
            zc[q] = make_zc((f16)(float)(nib + 1));
        }
        else
        {
            sc[q] = (f16)(float)((nib + 1) * (nib + 1)) * m.scale_src[g];
            zc[q] = make_zc((f16)(float)(1 << (BITS - 1)));
        }
    }
    f16x2 p[16];
    dequant_super<BITS>(lw.w, zc, p);
    #pragma unroll
    for (int q = 0; q < 4; q++)
    {
        if (q >= nvalid) continue;
        const f16x2 s2 = h2_dup(sc[q]);
        #pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const f16x2 v = p[4 * q + i] * s2;
            const int k = (chunk0 + q) * 32 + 8 * j + 2 * i;
            const int r0 = m.perm ? (int)m.perm[k] : k;            // packed row -> original row (q_matrix.cu:461)
            const int r1 = m.perm ? (int)m.perm[k + 1] : k + 1;
            out[(size_t)r0 * m.N + n] = v.x;
            out[(size_t)r1 * m.N + n] = v.y;
        }
    }
}

KERNEL void reconstruct_kernel(const QMatDev m, f16* out)
{
    const int tile = bid_x();
    const QDesc* dp = m.desc + bid_y();
    const int s = bid_z() * (nthreads() >> 6) + wave_id();
    if (s >= dp->n_super) return;
    const int lane = lane_id();
    if (m.is_gptq) { reconstruct_super<4, true>(m, dp, tile, s, lane, out); return; }
    switch (dp->bits)
    {
        case 2: reconstruct_super<2, false>(m, dp, tile, s, lane, out); break;
        case 3: reconstruct_super<3, false>(m, dp, tile, s, lane, out); break;
        case 4: reconstruct_super<4, false>(m, dp, tile, s, lane, out); break;
        case 5: reconstruct_super<5, false>(m, dp, tile, s, lane, out); break;
        case 6: reconstruct_super<6, false>(m, dp, tile, s, lane, out); break;
        case 8: reconstruct_super<8, false>(m, dp, tile, s, lane, out); break;
    }
}

// [tile][G][16] tables of the per-(group, column) scale (and GPTQ zero point) in the fp16 values reconstruct() uses
KERNEL void __launch_bounds__(256) scale_table_kernel(const QMatDev m, f16* sc_tab, f16* zp_tab)
{
    const int tile = bid_x();
    const int idx = bid_y() * 256 + tid();
    if (idx >= m.G * 16) return;
    const int g = idx >> 4, c = idx & 15;
    const int n = tile * TILE_N + c;
    const u32 word = m.q_scale[(size_t)g * (m.N >> 3) + (n >> 3)];
    const int nib = (word >> (4 * (n & 7))) & 15;
    const size_t o = ((size_t)tile * m.G + g) * 16 + c;
    if (m.is_gptq) { sc_tab[o] = m.scale_src[(size_t)g * m.N + n]; zp_tab[o] = (f16)(float)(nib + 1); }
    else           sc_tab[o] = (f16)(float)((nib + 1) * (nib + 1)) * m.scale_src[g];
}

// ---- host -----------------------------------------------------------------------------------------------------------

struct Section { int bits; int k0; int chunks; int qrow0; };

static bool bits_ok(int b) { return b == 2 || b == 3 || b == 4 || b == 5 || b == 6 || b == 8; }

void qmatrix_destroy(QMatrix* qm);

int qmatrix_create(QMatrix** out, int device, int K, int N, int G,
                   u32* q_weight, u16* q_perm, u16* q_invperm, u32* q_scale, f16* q_scale_max, u16* q_groups,
                   u32* gptq_qzeros, f16* gptq_scales, const u32* gptq_g_idx_host,
                   f16* bias, f16* temp_dq, int max_dq_rows, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    *out = nullptr;
    const bool is_gptq = gptq_qzeros != nullptr;
    EXL2_REQUIRE(q_weight != nullptr, "make_q_matrix: q_weight is null");
    EXL2_REQUIRE(N > 0 && N % TILE_N == 0, "make_q_matrix: width %d must be a multiple of %d", N, TILE_N);
    EXL2_REQUIRE(K > 0 && K % 32 == 0 && K < 65536, "make_q_matrix: height %d must be a multiple of 32 and < 65536", K);
    EXL2_REQUIRE(G > 0, "make_q_matrix: no groups");
    if (!is_gptq) EXL2_REQUIRE(q_scale && q_scale_max && q_groups, "make_q_matrix: EXL2 tensors missing");
    else          EXL2_REQUIRE(gptq_scales, "make_q_matrix: GPTQ scales missing");
    DeviceGuard on_device(device);

    const int n_chunks = K / 32;
    std::vector<u16> chunk_group(n_chunks);
    std::vector<Section> sections;
    std::vector<u16> x_map, x_map_inv;
    long long total_qrows = 0;
    int max_bits = 0;

    if (!is_gptq)
    {
        std::vector<u16> qg(2 * (size_t)G);
        HIP_TRY(hipMemcpyAsync(qg.data(), q_groups, qg.size() * sizeof(u16), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        // rows of every group from consecutive first-packed-row entries (q_matrix.cu:130-159); last group takes the rest
        int row = 0;
        for (int g = 0; g < G; g++)
        {
            const int bits = qg[2 * g];
            EXL2_REQUIRE(bits_ok(bits), "make_q_matrix: group %d has unsupported bit width %d", g, bits);
            int rows;
            if (g < G - 1)
            {
                const int qrows = (int)qg[2 * g + 3] - (int)qg[2 * g + 1];
                EXL2_REQUIRE(qrows > 0 && (qrows * 32) % bits == 0, "make_q_matrix: malformed q_groups at group %d", g);
                rows = qrows * 32 / bits;
            }
            else rows = K - row;
            EXL2_REQUIRE(rows > 0 && rows % 32 == 0 && row + rows <= K,
                         "make_q_matrix: group %d spans %d rows; this build needs multiples of 32", g, rows);
            EXL2_REQUIRE((int)qg[2 * g + 1] == (int)total_qrows, "make_q_matrix: q_groups rows not contiguous at group %d", g);
            if (sections.empty() || sections.back().bits != bits)
                sections.push_back({bits, row, 0, (int)qg[2 * g + 1]});
            sections.back().chunks += rows / 32;
            for (int c = 0; c < rows / 32; c++) chunk_group[row / 32 + c] = (u16)g;
            row += rows;
            total_qrows += (long long)rows * bits / 32;
            if (bits > max_bits) max_bits = bits;
        }
        EXL2_REQUIRE(row == K, "make_q_matrix: q_groups cover %d rows, height is %d", row, K);
    }
    else
    {
        int gs = 1;
        while (gs * G < K) gs *= 2;                                       // q_matrix.cu:101-105
        max_bits = 4;
        total_qrows = K / 8;
        sections.push_back({4, 0, n_chunks, 0});
        if (gptq_g_idx_host)
        {
            // stable counting sort of rows by group (q_matrix.cu:606-642)
            EXL2_REQUIRE(q_perm && q_invperm, "make_q_matrix: GPTQ act-order needs q_perm / q_invperm buffers");
            std::vector<u32> start(G + 1, 0);
            for (int r = 0; r < K; r++)
            {
                EXL2_REQUIRE(gptq_g_idx_host[r] < (u32)G, "make_q_matrix: g_idx[%d] out of range", r);
                start[gptq_g_idx_host[r] + 1]++;
            }
            for (int g = 0; g < G; g++) start[g + 1] += start[g];
            std::vector<u32> nxt(start.begin(), start.end() - 1);
            x_map.resize(K); x_map_inv.resize(K);
            for (int r = 0; r < K; r++) { const u32 t = nxt[gptq_g_idx_host[r]]++; x_map_inv[r] = (u16)t; x_map[t] = (u16)r; }
            for (int c = 0; c < n_chunks; c++)
            {
                const u32 g0 = gptq_g_idx_host[x_map[c * 32]], g1 = gptq_g_idx_host[x_map[c * 32 + 31]];
                EXL2_REQUIRE(g0 == g1, "make_q_matrix: GPTQ group boundaries must fall on multiples of 32 rows");
                chunk_group[c] = (u16)g0;
            }
        }
        else
        {
            EXL2_REQUIRE(gs % 32 == 0 || G == 1, "make_q_matrix: GPTQ group size %d not a multiple of 32", gs);
            for (int c = 0; c < n_chunks; c++) chunk_group[c] = (u16)((c * 32) / gs);
        }
    }

    // descriptors, ordered by K
    const int tiles = N / TILE_N;
    std::vector<QDesc> descs;
    std::vector<RelayoutArgs> jobs;
    size_t tail_words = 0;
    for (const Section& sec : sections)
    {
        const int F = sec.chunks / 4, tail = sec.chunks % 4;
        const u32 sec_base = (u32)((size_t)sec.qrow0 * N);
        const u32 stride = (u32)F * 64u * (u32)sec.bits;
        for (int s0 = 0; s0 < F; s0 += QDESC_MAX_SUPER)
        {
            QDesc d; memset(&d, 0, sizeof(d));
            d.n_super = (u16)((F - s0) < QDESC_MAX_SUPER ? (F - s0) : QDESC_MAX_SUPER);
            d.base_word = sec_base + (u32)s0 * 64u * (u32)sec.bits;
            d.tile_stride = stride;
            d.k_base = (u16)(sec.k0 + s0 * SUPER_ROWS);
            d.bits = (u8)sec.bits; d.nvalid_last = 4; d.in_tail = 0;
            descs.push_back(d);
            RelayoutArgs r; memset(&r, 0, sizeof(r));
            r.base_word = d.base_word; r.tile_stride = d.tile_stride; r.n_super = d.n_super; r.nvalid_last = 4;
            r.bits = sec.bits; r.is_gptq = is_gptq; r.N = N; r.k_base = d.k_base; r.src_qrow0 = sec.qrow0;
            r.chunk_in_sec0 = s0 * 4; r.in_tail = 0;
            jobs.push_back(r);
        }
        if (tail)
        {
            QDesc d; memset(&d, 0, sizeof(d));
            d.n_super = 1;
            d.base_word = (u32)tail_words;
            d.tile_stride = 64u * (u32)sec.bits;
            d.k_base = (u16)(sec.k0 + F * SUPER_ROWS);
            d.bits = (u8)sec.bits; d.nvalid_last = (u8)tail; d.in_tail = 1;
            descs.push_back(d);
            RelayoutArgs r; memset(&r, 0, sizeof(r));
            r.base_word = d.base_word; r.tile_stride = d.tile_stride; r.n_super = 1; r.nvalid_last = tail;
            r.bits = sec.bits; r.is_gptq = is_gptq; r.N = N; r.k_base = d.k_base; r.src_qrow0 = sec.qrow0;
            r.chunk_in_sec0 = F * 4; r.in_tail = 1;
            jobs.push_back(r);
            tail_words += (size_t)tiles * 64 * sec.bits;
        }
    }

    { u32 pre = 0; for (QDesc& d : descs) { d.sc_prefix = pre; pre += d.n_super; } }

    // everything allocated below is released on any early return (error paths included) unless `keep` is set at the end
    struct Hold
    {
        QMatrix* qm = nullptr; u32* temp = nullptr; bool keep = false;
        ~Hold() { if (temp) (void)hipFree(temp); if (qm && !keep) qmatrix_destroy(qm); }
    } hold;
    QMatrix* qm = (QMatrix*)calloc(1, sizeof(QMatrix));
    if (!qm) EXL2_FAIL(EXL2_E_OOM, "make_q_matrix: host out of memory");
    hold.qm = qm;
    qm->device = device; qm->height = K; qm->width = N; qm->groups = G; qm->is_gptq = is_gptq;
    qm->q_weight = q_weight; qm->q_perm = q_perm; qm->q_invperm = q_invperm;
    qm->temp_dq = temp_dq; qm->max_dq_rows = max_dq_rows; qm->max_bits = max_bits;
    qm->cg_host = (u16*)malloc((size_t)n_chunks * sizeof(u16));
    if (!qm->cg_host) EXL2_FAIL(EXL2_E_OOM, "make_q_matrix: host out of memory");
    memcpy(qm->cg_host, chunk_group.data(), (size_t)n_chunks * sizeof(u16));

    const size_t weight_words = (size_t)total_qrows * N;
    u32*& temp = hold.temp;
    hipError_t e = hipMalloc((void**)&temp, weight_words * sizeof(u32));
    if (e == hipSuccess && tail_words) e = hipMalloc((void**)&qm->tail_buf, tail_words * sizeof(u32));
    if (e == hipSuccess) e = hipMalloc((void**)&qm->desc_buf, descs.size() * sizeof(QDesc));
    chunk_group.resize(chunk_group.size() + 2, (u16)0);            // readable as whole dwords (LDS-DMA source)
    if (e == hipSuccess) e = hipMalloc((void**)&qm->chunk_group_buf, chunk_group.size() * sizeof(u16));
    if (e == hipSuccess && !is_gptq) e = hipMalloc((void**)&qm->scale_pad_buf, ((size_t)G + 2) * sizeof(f16));
    const size_t perm_bytes = q_perm ? (((size_t)K * 2 + 15) & ~(size_t)15) : 0;
    const size_t cg_bytes = ((size_t)n_chunks * 2 + 4 + 15) & ~(size_t)15;
    const size_t tab_elems = (size_t)(N / TILE_N) * G * 16;
    if (e == hipSuccess) e = hipMalloc((void**)&qm->pack_buf, perm_bytes + cg_bytes);
    // the chunk -> group map rides behind the scale table: the chained decode kernel derives its address from sc_tab
    if (e == hipSuccess) e = hipMalloc((void**)&qm->sc_tab_buf, tab_elems * sizeof(f16) + cg_bytes);
    if (e == hipSuccess && is_gptq) e = hipMalloc((void**)&qm->zp_tab_buf, tab_elems * sizeof(f16));
    if (e != hipSuccess)
    {
        (void)hipGetLastError();
        EXL2_FAIL(EXL2_E_OOM, "HIP out of memory (make_q_matrix: %zu bytes of re-layout scratch)", weight_words * 4);
    }
    HIP_TRY(hipMemcpyAsync(temp, q_weight, weight_words * sizeof(u32), hipMemcpyDeviceToDevice, stream));
    HIP_TRY(hipMemcpyAsync(qm->desc_buf, descs.data(), descs.size() * sizeof(QDesc), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(qm->chunk_group_buf, chunk_group.data(), chunk_group.size() * sizeof(u16), hipMemcpyHostToDevice, stream));
    if (qm->scale_pad_buf)
    {
        HIP_TRY(hipMemsetAsync(qm->scale_pad_buf, 0, ((size_t)G + 2) * sizeof(f16), stream));
        HIP_TRY(hipMemcpyAsync(qm->scale_pad_buf, q_scale_max, (size_t)G * sizeof(f16), hipMemcpyDeviceToDevice, stream));
    }
    if (!x_map.empty())
    {
        HIP_TRY(hipMemcpyAsync(q_perm, x_map.data(), K * sizeof(u16), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(q_invperm, x_map_inv.data(), K * sizeof(u16), hipMemcpyHostToDevice, stream));
    }
    for (RelayoutArgs& r : jobs)
    {
        r.src = temp;
        r.dst = r.in_tail ? qm->tail_buf : q_weight;
        r.row_map = (is_gptq && !x_map.empty()) ? q_perm : nullptr;
        dim3 grid((unsigned)tiles, (unsigned)((r.n_super + 3) / 4), 1);
        LAUNCH(relayout_kernel, grid, dim3(256, 1, 1), 0, stream, r);
    }
    // prologue pack (after q_perm has its final contents) and scale tables
    HIP_TRY(hipMemsetAsync(qm->pack_buf, 0, perm_bytes + cg_bytes, stream));
    if (q_perm) HIP_TRY(hipMemcpyAsync(qm->pack_buf, q_perm, (size_t)K * 2, hipMemcpyDeviceToDevice, stream));
    HIP_TRY(hipMemcpyAsync(qm->pack_buf + perm_bytes, chunk_group.data(), (size_t)n_chunks * 2, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemsetAsync((u8*)qm->sc_tab_buf + tab_elems * sizeof(f16), 0, cg_bytes, stream));
    HIP_TRY(hipMemcpyAsync((u8*)qm->sc_tab_buf + tab_elems * sizeof(f16), chunk_group.data(), (size_t)n_chunks * 2, hipMemcpyHostToDevice, stream));
    {
        QMatDev t;
        memset(&t, 0, sizeof(t));
        t.q_scale = is_gptq ? gptq_qzeros : q_scale; t.scale_src = is_gptq ? gptq_scales : q_scale_max;
        t.N = N; t.G = G; t.is_gptq = is_gptq ? 1 : 0;
        LAUNCH(scale_table_kernel, dim3((unsigned)(N / TILE_N), (unsigned)((G * 16 + 255) / 256), 1), dim3(256, 1, 1), 0, stream,
               t, qm->sc_tab_buf, qm->zp_tab_buf);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    { u32* t = temp; temp = nullptr; HIP_TRY(hipFree(t)); }

    QMatDev& d = qm->dev;
    d.pack = qm->pack_buf; d.pack_units = (u32)((perm_bytes + cg_bytes) / 16); d.pack_cg_off = (u32)perm_bytes;
    d.sc_tab = qm->sc_tab_buf; d.zp_tab = qm->zp_tab_buf;
    d.qw = q_weight; d.tail = qm->tail_buf; d.desc = qm->desc_buf; d.chunk_group = qm->chunk_group_buf;
    d.perm = q_perm;
    d.q_scale = is_gptq ? gptq_qzeros : q_scale;
    d.scale_src = is_gptq ? gptq_scales : q_scale_max;
    d.bias = bias;
    d.scale_pad = qm->scale_pad_buf;
    d.n_desc = (int)descs.size(); d.K = K; d.N = N; d.G = G; d.is_gptq = is_gptq ? 1 : 0;

    // runs for the streaming kernel: one per section (+ one per partial super-chunk), in K order
    {
        std::vector<QRun> runs;
        size_t tw = 0;
        for (const Section& sec : sections)
        {
            const int F = sec.chunks / 4, tail = sec.chunks % 4;
            if (F)
            {
                QRun r; memset(&r, 0, sizeof(r));
                r.base_word = (u32)((size_t)sec.qrow0 * N); r.tile_stride = (u32)F * 64u * (u32)sec.bits;
                r.n_super = (u16)F; r.k_base = (u16)sec.k0; r.bits = (u8)sec.bits; r.nvalid_last = 4; r.in_tail = 0;
                runs.push_back(r);
            }
            if (tail)
            {
                QRun r; memset(&r, 0, sizeof(r));
                r.base_word = (u32)tw; r.tile_stride = 64u * (u32)sec.bits; r.n_super = 1;
                r.k_base = (u16)(sec.k0 + F * SUPER_ROWS); r.bits = (u8)sec.bits; r.nvalid_last = (u8)tail; r.in_tail = 1;
                runs.push_back(r);
                tw += (size_t)tiles * 64 * sec.bits;
            }
        }
        d.n_runs = 0; d.main_run = 0;
        if (runs.size() <= MAX_RUNS)
        {
            d.n_runs = (int)runs.size();
            int best = -1;
            for (int i = 0; i < d.n_runs; i++)
            {
                d.runs[i] = runs[i];
                if (runs[i].nvalid_last == 4 && (best < 0 || runs[i].n_super > runs[best].n_super)) best = i;
            }
            d.main_run = best < 0 ? 0 : best;
        }
    }

    // algorithmic bytes of one pass over this matrix
    long long b = (long long)weight_words * 4;
    if (is_gptq) b += (long long)G * (N / 8) * 4 + (long long)G * N * 2;
    else         b += (long long)G * (N / 8) * 4 + (long long)G * 2 + (long long)K * 4;     // q_scale, q_scale_max, q_group_map
    if (q_perm) b += (long long)K * 2;
    qm->weight_bytes = b;

    hold.keep = true;
    *out = qm;
    return EXL2_OK;
}

void qmatrix_destroy(QMatrix* qm)
{
    if (!qm) return;
    DeviceGuard on_device(qm->device);
    if (qm->tail_buf) (void)hipFree(qm->tail_buf);
    if (qm->desc_buf) (void)hipFree(qm->desc_buf);
    if (qm->chunk_group_buf) (void)hipFree(qm->chunk_group_buf);
    if (qm->scale_pad_buf) (void)hipFree(qm->scale_pad_buf);
    if (qm->pack_buf) (void)hipFree(qm->pack_buf);
    if (qm->sc_tab_buf) (void)hipFree(qm->sc_tab_buf);
    if (qm->zp_tab_buf) (void)hipFree(qm->zp_tab_buf);
    free(qm->cg_host);
    free(qm);
}

int qmatrix_reconstruct(const QMatrix* qm, f16* out, void* stream)
{
    EXL2_REQUIRE(qm && out, "reconstruct: null argument");
    dim3 grid((unsigned)(qm->width / TILE_N), (unsigned)qm->dev.n_desc, (QDESC_MAX_SUPER + 3) / 4);
    LAUNCH(reconstruct_kernel, grid, dim3(256, 1, 1), 0, stream, qm->dev, out);
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}
