"""The reference's OWN decode kernels, executed on the host, composed into a model forward -- the measured yardstick for
model-level tolerances.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Needs oracle/_ref/*.so (oracle/ref_build/build.sh: the reference's kernel
text compiled for the host from where it lies under /root/reference; the libraries travel to the GPU box, the sources do not).

What runs: exactly the composition QAttn::forward_cuda_1 / _2 (exllamav2_ext/cuda/q_attn.cu:153-345) and QMLP::forward_run_
(q_mlp.cu:153-236) make for <= 4 rows -- rms_norm_kernel (rms_norm.cu:33-175) -> gemm_half_q_half_kernel (q_gemm_kernel.cuh:
140-565; GPTQ: q_gemm_kernel_gptq.cuh:61-246) x 3 -> rope (rope.cu:10-273) -> attention -> gemm into the residual stream
(clear = false: the split-K partial sums are added to the row one by one in fp16) -> rms_norm -> gemm x 2 -> act_mul_kernel
(q_mlp_activation.cuh:54-112) -> gemm into the residual stream; final norm + head gemm.  Attention itself is third party in
the reference (flash-attn / torch SDPA: no kernel text to execute), so it is the oracle's float64 attention here -- the SAME
function OracleModel uses, which keeps this model's distance from OracleModel a measure of the linear / norm / activation
kernels' fp16 behaviour alone.

Its distance from OracleModel (float64 accumulation, fp16 roundings where the reference stores fp16) is what "the reference
itself is this far from its own torch semantics" means; tests hold the HIP path to that distance, not to an asserted one
(tests/golden/make_golden_model_yardstick.py).
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import exl2 as OX
from . import modules as OM

F16 = np.float16
_REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")
vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float


def available() -> bool:
    return all(os.path.exists(os.path.join(_REF, f)) for f in
               ("libqmatrix_ref.so", "librmsnorm_ref.so", "librope_ref.so", "libactmul_ref.so"))


_libs = None


def _load():
    global _libs
    if _libs is None:
        qm = ctypes.CDLL(os.path.join(_REF, "libqmatrix_ref.so"))
        qm.ref_exl2_shuffle.argtypes = [vp, vp, ci, ci, ci]
        for f in (qm.ref_exl2_gemm, qm.ref_exl2_gemm_acc):
            f.argtypes = [vp, ci] + [vp] * 6 + [ci] * 4 + [vp]; f.restype = ci
        for f in (qm.ref_gptq_gemm, qm.ref_gptq_gemm_acc):
            f.argtypes = [vp, ci] + [vp] * 4 + [ci] * 3 + [vp]; f.restype = ci
        rn = ctypes.CDLL(os.path.join(_REF, "librmsnorm_ref.so"))
        rn.ref_rms_norm.argtypes = [vp, vp, vp, cf, ci, ci]; rn.ref_rms_norm.restype = ci
        rp = ctypes.CDLL(os.path.join(_REF, "librope_ref.so"))
        rp.ref_rope.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, vp, ci]; rp.ref_rope.restype = ci
        am = ctypes.CDLL(os.path.join(_REF, "libactmul_ref.so"))
        am.ref_act_mul.argtypes = [vp, vp, ci, ci, ci]; am.ref_act_mul.restype = ci
        _libs = (qm, rn, rp, am)
    return _libs


def _np(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


class _Exl2:
    """one EXL2 matrix as the reference holds it after load (shuffled in place, scale-max prescaled: ext.py:336, q_matrix.cu:189-195)"""

    def __init__(self, w: dict):
        qm = _load()[0]
        self.qw = np.ascontiguousarray(_np(w["q_weight"])).view(np.uint32).copy()
        r, self.n = self.qw.shape
        self.groups = _np(w["q_scale"]).shape[0]
        qg = _np(w["q_groups"])
        self.gmap = OX.make_group_map(qg, r).astype(np.uint16)
        self.k = self.gmap.shape[0] // 2
        self.smax = OX.exl2_prescale_scale_max(_np(w["q_scale_max"])).view(np.uint16).copy()
        self.perm = np.argsort(_np(w["q_invperm"]).astype(np.int64), kind="stable").astype(np.uint16)
        self.qs = np.ascontiguousarray(_np(w["q_scale"])).view(np.uint32).copy()
        self.qg = np.ascontiguousarray(qg).view(np.uint16).copy()
        assert qm.ref_exl2_shuffle(self.qw.ctypes.data, self.qg.ctypes.data, self.k, self.n, self.groups) == 0
        self.bias = _np(w["bias"]).astype(F16) if "bias" in w else None

    def gemm(self, a: np.ndarray, c: np.ndarray | None, block_kn: int) -> np.ndarray:
        """c = a . W (c None) or c += a . W (the residual stream), rows in blocks of <= 4 like blockIdx.y of the reference's launch"""
        qm = _load()[0]
        a = np.ascontiguousarray(a.astype(F16))
        acc = c is not None
        out = np.ascontiguousarray(c.astype(F16)).copy() if acc else np.full((a.shape[0], self.n), np.nan, dtype=F16)
        for r0 in range(0, a.shape[0], 4):
            ab, cb = np.ascontiguousarray(a[r0:r0 + 4]), np.ascontiguousarray(out[r0:r0 + 4])
            f = qm.ref_exl2_gemm_acc if acc else qm.ref_exl2_gemm
            assert f(ab.ctypes.data, ab.shape[0], self.qw.ctypes.data, self.perm.ctypes.data, self.qs.ctypes.data, self.smax.ctypes.data,
                     self.qg.ctypes.data, self.gmap.ctypes.data, self.k, self.n, self.groups, block_kn, cb.ctypes.data) == 0
            out[r0:r0 + 4] = cb
        return out


class _Gptq:
    def __init__(self, w: dict):
        self.qw = np.ascontiguousarray(_np(w["qweight"])).view(np.uint32).copy()
        self.k, self.n = self.qw.shape[0] * 8, self.qw.shape[1]
        self.groups = _np(w["qzeros"]).shape[0]
        g_idx = _np(w["g_idx"])
        sequential = np.array_equal(g_idx, np.arange(self.k) // (self.k // self.groups))
        self.perm = None if sequential else OX.gptq_sequential_perm(g_idx, self.groups)[0].astype(np.uint16)
        self.qz = np.ascontiguousarray(_np(w["qzeros"])).view(np.uint32).copy()
        self.sc = np.ascontiguousarray(_np(w["scales"]).astype(F16)).view(np.uint16).copy()
        self.bias = _np(w["bias"]).astype(F16) if "bias" in w else None

    def gemm(self, a: np.ndarray, c: np.ndarray | None, block_kn: int) -> np.ndarray:
        qm = _load()[0]
        a = np.ascontiguousarray(a.astype(F16))
        acc = c is not None
        out = np.ascontiguousarray(c.astype(F16)).copy() if acc else np.full((a.shape[0], self.n), np.nan, dtype=F16)
        for r0 in range(0, a.shape[0], 4):
            ab, cb = np.ascontiguousarray(a[r0:r0 + 4]), np.ascontiguousarray(out[r0:r0 + 4])
            f = qm.ref_gptq_gemm_acc if acc else qm.ref_gptq_gemm
            assert f(ab.ctypes.data, ab.shape[0], self.qw.ctypes.data, None if self.perm is None else self.perm.ctypes.data,
                     self.qz.ctypes.data, self.sc.ctypes.data, self.k, self.n, self.groups, cb.ctypes.data) == 0
            out[r0:r0 + 4] = cb
        return out


class ReferenceKernelModel:
    """Same call surface as OracleModel (reset / forward -> float64 logits); dense MLP, FP16 cache, <= 4 rows per kernel launch
    (more rows run as row blocks, which is what the reference's grid does).  block_kn: 32 or 64, the two block sizes the
    reference autotunes between (q_gemm.cu:100-107) -- they differ in where the split-K sums are rounded."""

    def __init__(self, cfg, ck: dict, block_kn: int = 32):
        assert available(), "oracle/_ref is not built (oracle/ref_build/build.sh)"
        self.cfg, self.block_kn = cfg, block_kn
        self.lin, self.t = {}, {}
        for k, v in ck.items():
            if isinstance(v, dict) and "qweight" in v:
                self.lin[k] = _Gptq(v)
            elif isinstance(v, dict):
                self.lin[k] = _Exl2(v)
            else:
                self.t[k] = _np(v).copy()
        self.sin, self.cos = OM.rope_tables(cfg.max_seq_len, cfg.head_dim, cfg.rotary_embedding_base, cfg.rope_style == 2)
        self.sin16, self.cos16 = np.ascontiguousarray(self.sin.astype(F16)), np.ascontiguousarray(self.cos.astype(F16))

    def reset(self, batch: int):
        c = self.cfg
        shape = (c.num_hidden_layers, batch, c.max_seq_len, c.num_key_value_heads, c.head_dim)
        self.k_cache, self.v_cache = np.zeros(shape, dtype=F16), np.zeros(shape, dtype=F16)
        self.seq_len = 0

    def _gemm(self, key, a, c=None):
        m = self.lin[key]
        y = m.gemm(a, c, self.block_kn)
        if m.bias is not None:
            y = (y.astype(np.float32) + m.bias.astype(np.float32)).astype(F16)
        return y

    def _norm(self, x, key):
        rn = _load()[1]
        x = np.ascontiguousarray(x.astype(F16)); w = np.ascontiguousarray(self.t[key].astype(F16))
        y = np.empty_like(x)
        assert rn.ref_rms_norm(x.ctypes.data, w.ctypes.data, y.ctypes.data, float(self.cfg.norm_eps), x.shape[0], x.shape[1]) == 0
        return y

    def _rope(self, x, heads, past, s):
        """x fp16 [b, s, heads, hd], rotated by the reference's rope kernel (rope.cu) in place"""
        rp = _load()[2]
        b = x.shape[0]
        x = np.ascontiguousarray(x.astype(F16))
        assert rp.ref_rope(x.ctypes.data, self.sin16.ctypes.data, self.cos16.ctypes.data, b, s * heads, self.cfg.head_dim, heads, past, None,
                           1 if self.cfg.rope_style == 2 else 0) == 0
        return x

    def forward(self, ids: np.ndarray) -> np.ndarray:
        am = _load()[3]
        c = self.cfg
        b, s = ids.shape
        past = self.seq_len
        x = self.t["model.embed_tokens"][ids.reshape(-1)].reshape(b * s, c.hidden_size).astype(F16)
        for i in range(c.num_hidden_layers):
            p = f"model.layers.{i}"
            n = self._norm(x, p + ".input_layernorm")
            q = self._gemm(p + ".self_attn.q_proj", n).reshape(b, s, c.num_attention_heads, c.head_dim)
            k = self._gemm(p + ".self_attn.k_proj", n).reshape(b, s, c.num_key_value_heads, c.head_dim)
            v = self._gemm(p + ".self_attn.v_proj", n).reshape(b, s, c.num_key_value_heads, c.head_dim)
            if c.rope_style:
                q = self._rope(q, c.num_attention_heads, past, s)
                k = self._rope(k, c.num_key_value_heads, past, s)
            self.k_cache[i, :, past:past + s] = k
            self.v_cache[i, :, past:past + s] = v
            a = OM.attention(q, self.k_cache[i, :, :past + s], self.v_cache[i, :, :past + s])
            a = a.reshape(b * s, c.num_attention_heads * c.head_dim)
            x = self._gemm(p + ".self_attn.o_proj", a, c=x)
            n = self._norm(x, p + ".post_attention_layernorm")
            g = np.ascontiguousarray(self._gemm(p + ".mlp.gate_proj", n))
            u = np.ascontiguousarray(self._gemm(p + ".mlp.up_proj", n))
            assert am.ref_act_mul(g.ctypes.data, u.ctypes.data, g.shape[0], g.shape[1], 0) == 0
            x = self._gemm(p + ".mlp.down_proj", g, c=x)
        self.seq_len = past + s
        n = self._norm(x, "model.norm")
        logits = self._gemm("lm_head", n).astype(np.float64)[:, :c.vocab_size]
        return logits.reshape(b, s, c.vocab_size)
