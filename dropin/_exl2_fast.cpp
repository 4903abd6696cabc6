// Compiled half of the drop-in `exllamav2_ext` (dropin/exllamav2_ext.py imports it when built): the per-token operator calls of the
// UNMODIFIED reference host -- q_attn_forward_1 / flash_attn_func / q_attn_forward_2 (attn.py:1128-1203) and q_mlp_forward_
// (mlp.py:318-366) -- bound with pybind11 over the C ABI of include/exl2_hip.h, the way the reference binds its own extension
// (ext_bindings.cpp:27-138: tensors in, raw pointers + current stream down).  Two things live here:
//
//   1. The binding itself: no ctypes, no per-argument Python (round 4 measured ~130 ctypes calls and ~35 k Python-level pointer
//      look-ups per 64 tokens under the reference's loop: profiles/history/r04_dropin_host_profile.txt).
//
//   2. The MODULE CHAIN behind the operator boundary (round-4 review, item 4).  The library's fast decode route is "chained": a
//      producer leaves the residual stream as (xp, ss) = (x times the consumer's RMSNorm weight in the consumer's act-order, partial
//      sums of squares) and the consumer's launch needs no prologue (exl2_q_attn_forward_1_chain, exl2_q_attn_forward_2_chain,
//      exl2_q_mlp_forward_chain).  Our own host (exllamav2_amd/model.py) wires that up explicitly.  The reference host cannot: it
//      calls module by module.  So the binding does it between the calls:
//        * it LEARNS the order of the modules from the calls themselves (module M finished on residual tensor x, then module N was
//          entered with the same x  ->  succ[M] = N);
//        * when M finishes and succ[M] is chain-capable, M's last launch also publishes (xp, ss) for N, and the binding remembers
//          {N, x's storage address, x's torch version counter, rows, stream} and keeps a reference to x (so the address cannot be
//          recycled for another tensor);
//        * when N is entered with an x that matches all of that, its launches start from the published hand-off; if ANYTHING differs
//          -- another module, another tensor, a torch in-place operation on x in between (the version counter moved), another row
//          count or stream -- N first publishes (xp, ss) from x itself with one small launch (exl2_publish_rows) and then runs the
//          same kernels.  Either way the result is the one the un-chained route computes (tests/test_dropin_fast.py).
//      flash_attn_func's output tensor gets a second, act-ordered copy in a buffer of the binding (exl2_attn_decode_fused_dual);
//      q_attn_forward_2 uses it when it is handed exactly that tensor (same address, same version).
//      What the version counter cannot see is a write through a raw pointer by some OTHER native library between two module calls
//      (the binding's own entry points that write activations call note_write()) -- and tensors made under torch.inference_mode(),
//      which is how the reference runs its forward (model.py:764), have no counter at all: see version_of() below for what the
//      binding relies on then, and EXL2_MODULE_CHAIN_VERIFY=1 for the self-check.  EXL2_MODULE_CHAIN=0 turns the chain off.
//
// The reference keeps a `graph_map` per module instead (q_attn.cu:171-200, q_mlp.cu:89-110): its modules are 5-9 launches each; on
// this route a module is one or two launches, so there is nothing for a per-module graph to amortise.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <c10/core/DeviceGuard.h>
#include <dlfcn.h>

#include <cstdlib>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <mutex>

namespace {

typedef const void* cvp;
struct Api
{
    const char* (*last_error)();
    int (*q_attn_forward_1)(void*, cvp, int, int, int, const int*, void*, void*, void*, cvp, cvp, int, void*);
    int (*q_attn_forward_2)(void*, void*, cvp, int, int, void*);
    int (*q_mlp_forward)(void*, void*, int, void*);
    int (*q_attn_chain_info)(void*, int*, cvp*, cvp*, cvp*);
    int (*q_mlp_chain_info)(void*, int*, cvp*, cvp*);
    int (*q_attn_forward_1_chain_rope)(void*, cvp, const float*, int, int, int, int, const int*, void*, void*, void*, cvp, cvp, void*);
    int (*q_attn_forward_2_chain)(void*, void*, cvp, int, cvp, cvp, void*, float*, int*, void*);
    int (*q_mlp_forward_chain)(void*, void*, cvp, const float*, int, int, cvp, cvp, void*, float*, int*, void*);
    int (*publish_rows)(cvp, int, int, cvp, cvp, void*, float*, void*);
    int (*attn_decode_fused_dual)(cvp, cvp, cvp, void*, void*, void*, cvp, cvp, const int*, const int*, int, int, int, int, int, int, int,
                                  int, float, int, int, int, void*, long long, void*, int, cvp, void*, void*);
    long long (*paged_attn_scratch_bytes)(int, int, int);
};
Api api;
void* dll = nullptr;
bool allow_cpu = false;

template <typename F> void bind(F& f, const char* name)
{
    void* p = dlsym(dll, name);
    if (!p) throw std::runtime_error(std::string("_exl2_fast: the library does not export ") + name);
    f = reinterpret_cast<F>(p);
}

void check(int rc)
{
    if (rc != 0) throw std::runtime_error(std::string(api.last_error ? api.last_error() : "libexl2_hip error"));
}

void* f16_ptr(const at::Tensor& t, const char* name)
{
    if (!t.defined() || t.is_meta()) return nullptr;
    if (t.scalar_type() != at::kHalf) throw std::runtime_error(std::string(name) + ": expected dtype torch.float16");
    if (!t.is_contiguous()) throw std::runtime_error(std::string(name) + ": tensor must be contiguous");
    if (!t.is_cuda() && !(allow_cpu && t.is_cpu())) throw std::runtime_error(std::string(name) + ": tensor must live on a HIP device; there is no CPU path");
    return t.data_ptr();
}
const int* i32_ptr(const at::Tensor& t, const char* name)
{
    if (!t.defined() || t.is_meta()) return nullptr;
    if (t.scalar_type() != at::kInt) throw std::runtime_error(std::string(name) + ": expected dtype torch.int32");
    if (!t.is_contiguous()) throw std::runtime_error(std::string(name) + ": tensor must be contiguous");
    return (const int*)t.data_ptr();
}
void* any_ptr(const at::Tensor& t) { return (!t.defined() || t.is_meta()) ? nullptr : t.data_ptr(); }
void* stream_of(const at::Tensor& t) { return t.is_cuda() ? (void*)c10::hip::getCurrentHIPStream(t.get_device()).stream() : nullptr; }

// ---- the module chain ----------------------------------------------------------------------------------------------------------

// torch's per-tensor version counter: every in-place torch operation on a tensor (or a view of it) moves it.  Tensors made under
// torch.inference_mode() -- the reference's forward is decorated with it (model.py:764) -- carry NO counter: for those the binding has
// the address, the row count, the stream and the kept reference only, i.e. it relies on what the reference's module loop guarantees
// (model.py:996-1020: nothing but module.forward and device moves touches x between two modules).  EXL2_MODULE_CHAIN_VERIFY=1 re-derives
// every hand-off from x on entry and compares (a debugging aid: it synchronises); EXL2_MODULE_CHAIN=0 turns the chain off.
int64_t version_of(const at::Tensor& t) { return t.is_inference() ? -2 : (int64_t)t._version(); }

struct ModInfo { bool attn, capable; cvp in_invperm, o_invperm, norm_w; int rows_ok = 16; };      // rows_ok: largest call the chained kernels took so far declined nothing up to
struct Identity
{
    at::Tensor keep;                    // holds the storage: the address below cannot be handed to another tensor meanwhile
    void* ptr = nullptr; int64_t version = -1; int rows = 0; void* stream = nullptr;
    void set(const at::Tensor& t, int r, void* s) { keep = t; ptr = t.data_ptr(); version = version_of(t); rows = r; stream = s; }
    void clear() { keep = at::Tensor(); ptr = nullptr; version = -1; rows = 0; stream = nullptr; }
    bool is(const at::Tensor& t, int r, void* s) const { return ptr && t.data_ptr() == ptr && version_of(t) == version && r == rows && s == stream; }
};
struct DevBuf { at::Tensor xp[2], ss[2], vxp, vss, packed, scratch, counters; int hidden = 0, attn_w = 0; long long scratch_bytes = 0; };

struct State
{
    bool on = true, verify = false; int max_rows = 16;
    // self-check sampling (default ON): the first `verify_first` uses of every consumer's published hand-off and every `verify_every`-th one
    // after that are re-derived from x and compared; a mismatch blocks that edge for good (the module then makes its own hand-off)
    int verify_first = 8, verify_every = 256;
    std::unordered_map<void*, long long> edge_uses;
    std::unordered_set<void*> blocked;
    long long n_mismatch = 0; bool warned = false;
    std::unordered_map<void*, ModInfo> info;
    std::unordered_map<void*, void*> succ;
    std::unordered_map<int, DevBuf> bufs;
    // the module that finished last, and on which residual tensor
    void* finisher = nullptr; Identity fin_x;
    // a published hand-off
    void* consumer = nullptr; Identity pub_x; int pub_buf = 0, pub_npart = 0, next_buf = 0;
    // attention: the module whose q_attn_forward_1 ran last, its q tensor; the output flash_attn_func returned and its packed copy
    void* cur_attn = nullptr; void* cur_q = nullptr; int cur_rows = 0;
    void* packed_for = nullptr; Identity attn_out; cvp packed_ptr = nullptr;
    long long n_chained = 0, n_published = 0, n_plain = 0, n_attn_fast = 0, n_verified = 0, n_declined = 0;
} S;
// one host thread at a time inside the binding (the GIL serialises Python callers already; this also covers callers that release it)
std::recursive_mutex g_mu;
#define LOCKED std::lock_guard<std::recursive_mutex> lock_(g_mu)

void drop_pending() { S.consumer = nullptr; S.pub_x.clear(); }
void drop_all() { drop_pending(); S.finisher = nullptr; S.fin_x.clear(); S.cur_attn = nullptr; S.cur_q = nullptr; S.packed_for = nullptr; S.attn_out.clear(); S.packed_ptr = nullptr; }

ModInfo& info_of(void* h, bool attn)
{
    auto it = S.info.find(h);
    if (it != S.info.end()) return it->second;
    ModInfo m; m.attn = attn; m.in_invperm = m.o_invperm = m.norm_w = nullptr;
    int cap = 0;
    if (attn) check(api.q_attn_chain_info(h, &cap, &m.in_invperm, &m.o_invperm, &m.norm_w));
    else check(api.q_mlp_chain_info(h, &cap, &m.in_invperm, &m.norm_w));
    m.capable = cap != 0;
    return S.info.emplace(h, m).first->second;
}

int dev_key(const at::Tensor& t) { return t.is_cuda() ? (int)t.get_device() : -1; }

DevBuf& bufs_for(const at::Tensor& x, int hidden)
{
    DevBuf& b = S.bufs[dev_key(x)];
    if (b.hidden != hidden)
    {
        auto h = x.options().dtype(at::kHalf), f = x.options().dtype(at::kFloat);
        for (int i = 0; i < 2; i++) { b.xp[i] = at::zeros({16, hidden}, h); b.ss[i] = at::zeros({16 * 512}, f); }
        b.vxp = at::zeros({16, hidden}, h); b.vss = at::zeros({16 * 512}, f);
        b.hidden = hidden;
    }
    return b;
}

// on entry of module `h` with residual tensor x: learn the order, return the hand-off to start from (published by the predecessor,
// or made here from x)
struct HandOff { cvp xp; const float* ss; int npart; };
HandOff enter(void* h, const ModInfo& mi, const at::Tensor& x, int rows, int hidden, void* stream)
{
    if (S.finisher && S.fin_x.is(x, rows, stream)) S.succ[S.finisher] = h;
    DevBuf& b = bufs_for(x, hidden);
    HandOff ho;
    if (S.consumer == h && S.pub_x.is(x, rows, stream))
    {
        ho.xp = b.xp[S.pub_buf].data_ptr(); ho.ss = (const float*)b.ss[S.pub_buf].data_ptr(); ho.npart = S.pub_npart;
        S.n_chained++;
        const long long uses = S.edge_uses[h]++;
        if (S.verify || uses < S.verify_first || (S.verify_every > 0 && uses % S.verify_every == 0))
        {
            // what this module would have published from x itself must be what its predecessor left (same fp32 product, same rounding)
            check(api.publish_rows(x.data_ptr(), rows, hidden, mi.in_invperm, mi.norm_w, b.vxp.data_ptr(), (float*)b.vss.data_ptr(), stream));
            S.n_verified++;
            if (!at::equal(b.vxp.narrow(0, 0, rows), b.xp[S.pub_buf].narrow(0, 0, rows)))
            {
                if (S.verify)
                    throw std::runtime_error("_exl2_fast: EXL2_MODULE_CHAIN_VERIFY: the residual tensor changed between two module calls "
                                             "(the published hand-off no longer matches it)");
                // somebody wrote x between two module calls without the binding seeing it (a raw-pointer write without
                // note_write, or an in-place torch op on a tensor without a version counter): this edge is never chained again, and this
                // call starts from the hand-off just re-derived from x
                S.n_mismatch++; S.blocked.insert(h);
                if (!S.warned) { S.warned = true; fprintf(stderr, "[exl2] module chain: a residual tensor changed between two module calls; that hand-off is disabled (EXL2_MODULE_CHAIN_VERIFY=1 raises instead)\n"); }
                ho.xp = b.vxp.data_ptr(); ho.ss = (const float*)b.vss.data_ptr(); ho.npart = 1;
                S.n_chained--; S.n_published++;
            }
        }
    }
    else
    {
        const int k = S.next_buf; S.next_buf ^= 1;
        check(api.publish_rows(x.data_ptr(), rows, hidden, mi.in_invperm, mi.norm_w, b.xp[k].data_ptr(), (float*)b.ss[k].data_ptr(), stream));
        ho.xp = b.xp[k].data_ptr(); ho.ss = (const float*)b.ss[k].data_ptr(); ho.npart = 1;
        S.n_published++;
    }
    drop_pending();
    S.finisher = nullptr; S.fin_x.clear();
    return ho;
}

// what module `h`'s last launch should publish, and for whom
struct Publish { void* next; cvp invperm, norm_w; void* xp; float* ss; int buf; };
Publish successor(void* h, const at::Tensor& x, int hidden)
{
    Publish p{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    auto it = S.succ.find(h);
    if (it == S.succ.end()) return p;
    auto in = S.info.find(it->second);
    if (in == S.info.end() || !in->second.capable || S.blocked.count(it->second)) return p;
    DevBuf& b = bufs_for(x, hidden);
    p.next = it->second; p.invperm = in->second.in_invperm; p.norm_w = in->second.norm_w;
    p.buf = S.next_buf; S.next_buf ^= 1;
    p.xp = b.xp[p.buf].data_ptr(); p.ss = (float*)b.ss[p.buf].data_ptr();
    return p;
}
void finished(void* h, const at::Tensor& x, int rows, void* stream, const Publish& p, int npart)
{
    S.finisher = h; S.fin_x.set(x, rows, stream);
    if (p.next) { S.consumer = p.next; S.pub_x.set(x, rows, stream); S.pub_buf = p.buf; S.pub_npart = npart; }
    else drop_pending();
}

bool chain_rows(int rows) { return S.on && rows >= 1 && rows <= S.max_rows; }

// ---- bindings ----------------------------------------------------------------------------------------------------------------

void init(const std::string& path, bool cpu_ok)
{
    if (dll) { dlclose(dll); dll = nullptr; }
    dll = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);      // (LOCAL: a process may hold the emulation twin and the product library side by side -- the test-suite does)
    if (!dll) throw std::runtime_error(std::string("_exl2_fast: cannot load ") + path + ": " + dlerror() + " (there is no CPU fallback)");
    allow_cpu = cpu_ok;
    bind(api.last_error, "exl2_last_error");
    bind(api.q_attn_forward_1, "exl2_q_attn_forward_1");
    bind(api.q_attn_forward_2, "exl2_q_attn_forward_2");
    bind(api.q_mlp_forward, "exl2_q_mlp_forward");
    bind(api.q_attn_chain_info, "exl2_q_attn_chain_info");
    bind(api.q_mlp_chain_info, "exl2_q_mlp_chain_info");
    bind(api.q_attn_forward_1_chain_rope, "exl2_q_attn_forward_1_chain_rope");
    bind(api.q_attn_forward_2_chain, "exl2_q_attn_forward_2_chain");
    bind(api.q_mlp_forward_chain, "exl2_q_mlp_forward_chain");
    bind(api.publish_rows, "exl2_publish_rows");
    bind(api.attn_decode_fused_dual, "exl2_attn_decode_fused_dual");
    bind(api.paged_attn_scratch_bytes, "exl2_paged_attn_scratch_bytes");
    S = State();
    if (const char* e = getenv("EXL2_MODULE_CHAIN")) S.on = atoi(e) != 0;
    if (const char* e = getenv("EXL2_MODULE_CHAIN_VERIFY")) S.verify = atoi(e) != 0;
    if (const char* e = getenv("EXL2_MODULE_CHAIN_VERIFY_FIRST")) S.verify_first = atoi(e) < 0 ? 0 : atoi(e);
    if (const char* e = getenv("EXL2_MODULE_CHAIN_VERIFY_EVERY")) S.verify_every = atoi(e) < 0 ? 0 : atoi(e);
    if (const char* e = getenv("EXL2_MODULE_CHAIN_ROWS")) { const int v = atoi(e); if (v >= 1 && v <= 16) S.max_rows = v; }
}

void q_attn_forward_1(uintptr_t handle, const at::Tensor& x, int batch_size, int q_len, int past_len, const at::Tensor& past_lens,
                      const at::Tensor& q_temp, const at::Tensor& k_temp, const at::Tensor& v_temp, const at::Tensor& sin,
                      const at::Tensor& cos, const py::object& loras, const py::object& loras_temp)
{
    LOCKED;
    if (!loras.is_none() && py::len(loras) > 0) throw std::runtime_error("q_attn_forward_1: LoRA is out of scope of this build");
    void* h = (void*)handle;
    c10::OptionalDeviceGuard guard; if (x.is_cuda()) guard.reset_device(x.device());
    void* stream = stream_of(x);
    void* xq = f16_ptr(x, "x");
    void* q = f16_ptr(q_temp, "q_temp"); void* k = f16_ptr(k_temp, "k_temp"); void* v = f16_ptr(v_temp, "v_temp");
    const int rows = batch_size * q_len;
    const int hidden = (int)x.size(-1);
    S.cur_attn = nullptr; S.cur_q = nullptr; S.packed_for = nullptr;
    if (chain_rows(rows))
    {
        ModInfo& mi = info_of(h, true);
        if (mi.capable && rows <= mi.rows_ok)
        {
            const HandOff ho = enter(h, mi, x, rows, hidden, stream);
            const int rc = api.q_attn_forward_1_chain_rope(h, ho.xp, ho.ss, ho.npart, batch_size, q_len, past_len, i32_ptr(past_lens, "past_lens"),
                                                           q, k, v, any_ptr(sin), any_ptr(cos), stream);
            if (rc == 0) { S.cur_attn = h; S.cur_q = q; S.cur_rows = rows; return; }
            // a shape the chained kernels do not cover at this row count: nothing was launched (include/exl2_hip.h) -- the plain entry
            // point takes this call and every later one of this size
            mi.rows_ok = rows - 1; S.n_declined++;
        }
    }
    drop_all();
    S.n_plain++;
    check(api.q_attn_forward_1(h, xq, batch_size, q_len, past_len, i32_ptr(past_lens, "past_lens"), q, k, v, any_ptr(sin), any_ptr(cos), 1, stream));
}

// flash_attn_func as the reference calls it for a decode step with its K/V rows already in the cache (attn.py:960-977 behind
// attn.py:1088-1091, 1166-1173): q [1, s, H, hd], k / v [1, n, KVH, hd] contiguous, causal.  Returns None when the call is not of
// that shape (the Python shim then takes the general route).
py::object flash_attn_decode(const at::Tensor& q, const at::Tensor& k, const at::Tensor& v, double softmax_scale)
{
    LOCKED;
    if (q.dim() != 4 || k.dim() != 4 || v.dim() != 4 || q.size(0) != 1 || k.size(0) != 1 || !q.is_contiguous() || !k.is_contiguous() || !v.is_contiguous()
        || q.scalar_type() != at::kHalf || k.scalar_type() != at::kHalf || v.scalar_type() != at::kHalf)
        return py::none();
    const int s = (int)q.size(1), H = (int)q.size(2), hd = (int)q.size(3), n = (int)k.size(1), KVH = (int)k.size(2);
    if (KVH <= 0 || H % KVH || s < 1 || n < s || s * (H / KVH) > 32 || !(hd == 64 || hd == 128 || hd == 256)) return py::none();
    if (k.size(3) != hd || v.size(1) != n || v.size(2) != KVH || v.size(3) != hd) return py::none();
    if (!q.is_cuda() && !allow_cpu) return py::none();
    c10::OptionalDeviceGuard guard; if (q.is_cuda()) guard.reset_device(q.device());
    void* stream = stream_of(q);
    DevBuf& b = S.bufs[dev_key(q)];
    const long long need = api.paged_attn_scratch_bytes(s * H, hd, 64);
    if (b.scratch_bytes < need) { b.scratch = at::zeros({(need + 3) / 4}, q.options().dtype(at::kFloat)); b.scratch_bytes = need; }
    if (!b.counters.defined()) b.counters = at::zeros({4096}, q.options().dtype(at::kInt));
    if (b.attn_w != H * hd) { b.packed = at::zeros({32, (long)H * hd}, q.options()); b.attn_w = H * hd; }
    at::Tensor out = at::empty_like(q);
    cvp invperm = nullptr;
    bool chained = false;
    if (S.cur_attn && S.cur_q == q.data_ptr() && S.cur_rows == s)
    {
        auto it = S.info.find(S.cur_attn);
        if (it != S.info.end() && it->second.capable) { invperm = it->second.o_invperm; chained = true; }
    }
    // the step's own rows sit at positions n - s .. n - 1 of k / v already (rotated by q_attn_forward_1): they are "the new keys" of
    // the fused launch -- read from there, used, and stored back where they came from -- and nothing is rotated here
    const size_t row = (size_t)KVH * hd;
    const at::Half* kp = (const at::Half*)k.data_ptr(); const at::Half* vp = (const at::Half*)v.data_ptr();
    const int rc = api.attn_decode_fused_dual(q.data_ptr(), kp + (size_t)(n - s) * row, vp + (size_t)(n - s) * row, (void*)kp, (void*)vp,
                                              b.packed.data_ptr(), nullptr, nullptr, nullptr, nullptr, 1, s, H, KVH, hd, n, 0, n - s,
                                              (float)softmax_scale, 0, 0, 0, b.scratch.data_ptr(), b.scratch_bytes, b.counters.data_ptr(), 4096,
                                              chained ? invperm : nullptr, out.data_ptr(), stream);
    if (rc == 1) return py::none();                      // shape not covered by the one-launch kernel: nothing was launched
    check(rc);
    S.n_attn_fast++;
    if (chained)
    {
        S.packed_for = S.cur_attn; S.attn_out.set(out, s, stream);
        S.packed_ptr = invperm ? (cvp)b.packed.data_ptr() : (cvp)out.data_ptr();
    }
    return py::cast(out);
}

// flash_attn_with_kvcache as the reference's paged mode calls it for decode-sized steps (attn.py:602-613: the dynamic generator):
// q [b, s, H, hd], k / v [b, s, KVH, hd] (rotated by q_attn_forward_1) appended at cache_seqlens through block_table, then attention
// over the pages.  One launch instead of append + split attention + merge; None = not this shape (the Python shim's general route).
py::object flash_attn_kvcache_decode(const at::Tensor& q, const at::Tensor& k_cache, const at::Tensor& v_cache, const at::Tensor& k,
                                     const at::Tensor& v, const at::Tensor& cache_seqlens, const at::Tensor& block_table, double softmax_scale)
{
    LOCKED;
    for (const at::Tensor* t : {&q, &k_cache, &v_cache, &k, &v})
        if (!t->defined() || t->dim() != 4 || !t->is_contiguous() || t->scalar_type() != at::kHalf) return py::none();
    if (!cache_seqlens.defined() || !block_table.defined() || cache_seqlens.scalar_type() != at::kInt || block_table.scalar_type() != at::kInt
        || !cache_seqlens.is_contiguous() || !block_table.is_contiguous() || block_table.dim() != 2)
        return py::none();
    const int bsz = (int)q.size(0), s = (int)q.size(1), H = (int)q.size(2), hd = (int)q.size(3), KVH = (int)k.size(2);
    const int page_size = (int)k_cache.size(1), pps = (int)block_table.size(1);
    if (KVH <= 0 || H % KVH || s < 1 || s * (H / KVH) > 32 || !(hd == 64 || hd == 128 || hd == 256) || (page_size & (page_size - 1))) return py::none();
    if (k.size(0) != bsz || k.size(1) != s || k.size(3) != hd || v.sizes() != k.sizes() || k_cache.size(2) != KVH || k_cache.size(3) != hd
        || v_cache.sizes() != k_cache.sizes() || cache_seqlens.numel() != bsz || block_table.size(0) != bsz || bsz * s > 32)
        return py::none();
    if (!q.is_cuda() && !allow_cpu) return py::none();
    c10::OptionalDeviceGuard guard; if (q.is_cuda()) guard.reset_device(q.device());
    void* stream = stream_of(q);
    DevBuf& b = S.bufs[dev_key(q)];
    const long long need = api.paged_attn_scratch_bytes(bsz * s * H, hd, 64);
    if (b.scratch_bytes < need) { b.scratch = at::zeros({(need + 3) / 4}, q.options().dtype(at::kFloat)); b.scratch_bytes = need; }
    if (!b.counters.defined()) b.counters = at::zeros({4096}, q.options().dtype(at::kInt));
    if (b.attn_w != H * hd) { b.packed = at::zeros({32, (long)H * hd}, q.options()); b.attn_w = H * hd; }
    at::Tensor out = at::empty_like(q);
    cvp invperm = nullptr;
    bool chained = false;
    if (S.cur_attn && S.cur_q == q.data_ptr() && S.cur_rows == bsz * s)
    {
        auto it = S.info.find(S.cur_attn);
        if (it != S.info.end() && it->second.capable) { invperm = it->second.o_invperm; chained = true; }
    }
    const int rc = api.attn_decode_fused_dual(q.data_ptr(), k.data_ptr(), v.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), b.packed.data_ptr(),
                                              nullptr, nullptr, (const int*)cache_seqlens.data_ptr(), (const int*)block_table.data_ptr(),
                                              bsz, s, H, KVH, hd, page_size, pps, 0, (float)softmax_scale, 0, 0, 0, b.scratch.data_ptr(),
                                              b.scratch_bytes, b.counters.data_ptr(), 4096, chained ? invperm : nullptr, out.data_ptr(), stream);
    if (rc == 1) return py::none();
    check(rc);
    S.n_attn_fast++;
    if (chained)
    {
        S.packed_for = S.cur_attn; S.attn_out.set(out, bsz * s, stream);
        S.packed_ptr = invperm ? (cvp)b.packed.data_ptr() : (cvp)out.data_ptr();
    }
    return py::cast(out);
}

void q_attn_forward_2(uintptr_t handle, const at::Tensor& x, const at::Tensor& attn_output, int batch_size, int q_len,
                      const py::object& loras, const py::object& loras_temp)
{
    LOCKED;
    if (!loras.is_none() && py::len(loras) > 0) throw std::runtime_error("q_attn_forward_2: LoRA is out of scope of this build");
    void* h = (void*)handle;
    c10::OptionalDeviceGuard guard; if (x.is_cuda()) guard.reset_device(x.device());
    void* stream = stream_of(x);
    void* xq = f16_ptr(x, "x"); void* ao = f16_ptr(attn_output, "attn_output");
    const int rows = batch_size * q_len;
    const int hidden = (int)x.size(-1);
    if (chain_rows(rows) && S.packed_for == h && S.attn_out.is(attn_output, rows, stream))
    {
        const Publish p = successor(h, x, hidden);
        int npart = 0;
        const int rc = api.q_attn_forward_2_chain(h, xq, S.packed_ptr, rows, p.invperm, p.norm_w, p.xp, p.ss, &npart, stream);
        S.packed_for = nullptr; S.attn_out.clear();
        if (rc == 0)
        {
            S.n_chained++;
            finished(h, x, rows, stream, p, npart);
            return;
        }
        info_of(h, true).rows_ok = rows - 1; S.n_declined++;       // (nothing was launched: the plain entry point below takes the call)
    }
    S.packed_for = nullptr; S.attn_out.clear();
    S.n_plain++;
    check(api.q_attn_forward_2(h, xq, ao, batch_size, q_len, stream));
    // (the order is still learnt: the next module publishes its own hand-off)
    if (chain_rows(rows)) { Publish none{nullptr, nullptr, nullptr, nullptr, nullptr, 0}; finished(h, x, rows, stream, none, 0); }
    else drop_all();
}

void q_mlp_forward_(uintptr_t handle, const at::Tensor& x, const py::object& loras, const py::object& loras_temp)
{
    LOCKED;
    if (!loras.is_none() && py::len(loras) > 0) throw std::runtime_error("q_mlp_forward_: LoRA is out of scope of this build");
    void* h = (void*)handle;
    c10::OptionalDeviceGuard guard; if (x.is_cuda()) guard.reset_device(x.device());
    void* stream = stream_of(x);
    void* xq = f16_ptr(x, "x");
    const int hidden = (int)x.size(-1);
    const int rows = (int)(x.numel() / hidden);
    if (chain_rows(rows))
    {
        ModInfo& mi = info_of(h, false);
        if (mi.capable && rows <= mi.rows_ok)
        {
            const HandOff ho = enter(h, mi, x, rows, hidden, stream);
            const Publish p = successor(h, x, hidden);
            int npart = 0;
            const int rc = api.q_mlp_forward_chain(h, xq, ho.xp, ho.ss, ho.npart, rows, p.invperm, p.norm_w, p.xp, p.ss, &npart, stream);
            if (rc == 0) { finished(h, x, rows, stream, p, npart); return; }
            // declined (e.g. a down_proj too deep for one register pass at this row count): at most gate | up ran, into the module's
            // scratch -- x is untouched, the plain entry point recomputes the block
            mi.rows_ok = rows - 1; S.n_declined++;
        }
    }
    drop_all();
    S.n_plain++;
    check(api.q_mlp_forward(h, xq, rows, stream));
}

// an entry point of the Python half wrote `t` through a raw pointer (rms_norm_, gemm_half_q_half's c, ...): a hand-off published for
// that address is stale
void note_write(const at::Tensor& t)
{
    LOCKED;
    if (!t.defined() || t.is_meta()) return;
    void* p = t.data_ptr();
    if (p == S.pub_x.ptr) drop_pending();
    if (p == S.fin_x.ptr) { S.finisher = nullptr; S.fin_x.clear(); }
    if (p == S.attn_out.ptr) { S.packed_for = nullptr; S.attn_out.clear(); }
}

void forget_module(uintptr_t handle)
{
    LOCKED;
    void* h = (void*)handle;
    S.info.erase(h); S.succ.erase(h); S.edge_uses.erase(h); S.blocked.erase(h);
    for (auto it = S.succ.begin(); it != S.succ.end();) { if (it->second == h) it = S.succ.erase(it); else ++it; }
    drop_all();
}

py::dict stats(bool reset)
{
    LOCKED;
    py::dict d;
    d["chained"] = S.n_chained; d["published"] = S.n_published; d["plain"] = S.n_plain; d["attn_fast"] = S.n_attn_fast;
    d["verified"] = S.n_verified; d["declined"] = S.n_declined; d["on"] = S.on; d["max_rows"] = S.max_rows; d["known_successors"] = (long long)S.succ.size();
    d["verify_mismatch"] = S.n_mismatch; d["blocked_edges"] = (long long)S.blocked.size(); d["verify_first"] = S.verify_first; d["verify_every"] = S.verify_every;
    if (reset) { S.n_chained = S.n_published = S.n_plain = S.n_attn_fast = S.n_verified = S.n_declined = S.n_mismatch = 0; }
    return d;
}

void set_chain(bool on) { S.on = on; drop_all(); }
void set_verify(bool on) { S.verify = on; }
void set_verify_sampling(int first, int every) { S.verify_first = first < 0 ? 0 : first; S.verify_every = every < 0 ? 0 : every; }
void set_max_rows(int n) { if (n >= 1 && n <= 16) S.max_rows = n; drop_all(); }
void reset() { drop_all(); S.succ.clear(); S.info.clear(); S.bufs.clear(); S.edge_uses.clear(); S.blocked.clear(); S.warned = false; }

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "compiled half of the exllamav2_ext drop-in over libexl2_hip.so (see dropin/_exl2_fast.cpp)";
    m.def("init", &init, py::arg("library_path"), py::arg("allow_cpu") = false);
    m.def("q_attn_forward_1", &q_attn_forward_1);
    m.def("q_attn_forward_2", &q_attn_forward_2);
    m.def("q_mlp_forward_", &q_mlp_forward_);
    m.def("flash_attn_decode", &flash_attn_decode);
    m.def("flash_attn_kvcache_decode", &flash_attn_kvcache_decode);
    m.def("note_write", &note_write);
    m.def("forget_module", &forget_module);
    m.def("stats", &stats, py::arg("reset") = false);
    m.def("set_chain", &set_chain);
    m.def("set_verify", &set_verify);
    m.def("set_verify_sampling", &set_verify_sampling, py::arg("first"), py::arg("every"));
    m.def("set_max_rows", &set_max_rows);
    m.def("reset", &reset);
}
