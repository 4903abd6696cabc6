// abi_qmatrix.hip -- C ABI: q_matrix handles, reconstruct, gemm_half_q_half, make_group_map (include/exl2_hip.h).
#include "qmatrix.h"
#include "errors.h"
#include <string.h>

int qmatrix_create(QMatrix** out, int device, int K, int N, int G,
                   u32* q_weight, u16* q_perm, u16* q_invperm, u32* q_scale, f16* q_scale_max, u16* q_groups,
                   u32* gptq_qzeros, f16* gptq_scales, const u32* gptq_g_idx_host,
                   f16* bias, f16* temp_dq, int max_dq_rows, void* stream);
void qmatrix_destroy(QMatrix* qm);
int qmatrix_reconstruct(const QMatrix* qm, f16* out, void* stream);
int qgemv_launch(GemvJob* jobs, int n_jobs, int M, bool gptq, void* stream);

static thread_local char g_err[512] = "";

void exl2_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

const char* exl2_last_error(void) { return g_err; }

int exl2_abi_version(void) { return 1; }

int exl2_make_q_matrix(void** handle, int device, int height, int width, int groups,
                       void* q_weight, void* q_perm, void* q_invperm, void* q_scale, void* q_scale_max, void* q_groups,
                       void* gptq_qzeros, void* gptq_scales, const uint32_t* gptq_g_idx_host,
                       void* bias, void* temp_dq, int max_dq_rows, void* stream)
{
    EXL2_REQUIRE(handle, "make_q_matrix: null handle pointer");
    QMatrix* qm = nullptr;
    const int rc = qmatrix_create(&qm, device, height, width, groups, (u32*)q_weight, (u16*)q_perm, (u16*)q_invperm,
                                  (u32*)q_scale, (f16*)q_scale_max, (u16*)q_groups, (u32*)gptq_qzeros, (f16*)gptq_scales,
                                  gptq_g_idx_host, (f16*)bias, (f16*)temp_dq, max_dq_rows, stream);
    *handle = qm;
    return rc;
}

int exl2_free_q_matrix(void* handle)
{
    qmatrix_destroy((QMatrix*)handle);
    return EXL2_OK;
}

int exl2_q_matrix_info(void* handle, int* height, int* width, int* groups, int* is_gptq, long long* weight_bytes)
{
    EXL2_REQUIRE(handle, "q_matrix_info: null handle");
    const QMatrix* qm = (const QMatrix*)handle;
    if (height) *height = qm->height;
    if (width) *width = qm->width;
    if (groups) *groups = qm->groups;
    if (is_gptq) *is_gptq = qm->is_gptq ? 1 : 0;
    if (weight_bytes) *weight_bytes = qm->weight_bytes;
    return EXL2_OK;
}

int exl2_reconstruct(void* handle, void* out, void* stream)
{
    return qmatrix_reconstruct((const QMatrix*)handle, (f16*)out, stream);
}

// c[M, N] (+)= a[M, K] * W   (gemm_half_q_half_cuda, q_gemm.cu:201-313).  clear = 0 accumulates into c.
int exl2_gemm_half_q_half(const void* a, void* handle, void* c, int size_m, int clear,
                          const void* r_weights, int r_weights_stride, int mul_r_weights, void* stream)
{
    EXL2_REQUIRE(handle && a && c, "gemm_half_q_half: null argument");
    if (size_m <= 0) return EXL2_OK;
    const QMatrix* qm = (const QMatrix*)handle;
    GemvJob j;
    memset(&j, 0, sizeof(j));
    j.m = qm->dev;
    j.a = (const f16*)a; j.c = (f16*)c;
    j.lda = qm->height; j.ldc = qm->width;
    j.a_mode = A_PLAIN; j.c_mode = clear ? C_STORE : C_ACCUM;
    j.r_weights = (const f16*)r_weights; j.r_stride = r_weights_stride; j.mul_r_weights = mul_r_weights;
    const int rc = qgemv_launch(&j, 1, size_m, qm->is_gptq, stream);
    if (rc != 0) EXL2_FAIL(EXL2_E_INVALID, "gemm_half_q_half: launch configuration rejected (%d)", rc);
    HIP_TRY(hipGetLastError());
    return EXL2_OK;
}

// ext_qmatrix.cpp:341-361 / ext.py:301-316 -- host only
int exl2_make_group_map(const uint16_t* q_groups_host, int groups, int num_qrows, uint16_t* out, int out_len)
{
    EXL2_REQUIRE(q_groups_host && out, "make_group_map: null argument");
    int pos = 0;
    for (int i = 0; i < groups; i++)
    {
        const int bits = q_groups_host[2 * i];
        EXL2_REQUIRE(bits > 0, "make_group_map: zero bit width in group %d", i);
        const int q0 = q_groups_host[2 * i + 1];
        const int q1 = (i < groups - 1) ? q_groups_host[2 * i + 3] : num_qrows;
        const int rows = (q1 - q0) * 32 / bits;
        for (int r = 0; r < rows; r++)
        {
            EXL2_REQUIRE(pos + 2 <= out_len, "make_group_map: output too small");
            out[pos++] = (uint16_t)i;
            out[pos++] = (uint16_t)(rows - r);
        }
    }
    return pos;
}

}  // extern "C"
