"""Host-side mirror of the reference's operator boundary `ext_c.*` (exllamav2/ext.py:291, the pybind module
`exllamav2_ext`, ext_bindings.cpp:27-138) on top of the C-ABI HIP library.

Same names, argument order and meaning as the reference bindings; tensors are torch tensors, handles are python ints,
"None" tensors are passed as the meta-device sentinel `none_tensor` (ext.py:296).  Argument errors raise RuntimeError
like TORCH_CHECK does.  PyTorch is plumbing here: it owns device memory and streams; all arithmetic on this path runs
in the hand-written gfx950 kernels.
"""
from __future__ import annotations

import ctypes as C
import functools
import os

import torch

from . import _lib
from ._lib import Exl2Error

none_tensor = torch.empty((1, 1), device="meta")


def _is_none(t) -> bool:
    return t is None or (isinstance(t, torch.Tensor) and t.is_meta)


# (the host layer sits on the per-token path of the unmodified reference host -- ~130 calls x ~10 tensors per decoded token,
# tools/dropin_decode_bench.py -- so the argument checks below read tensor attributes that cost no Python-level object)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


class ExtC:
    """Operator surface bound to one C-ABI library.  `exllamav2_amd.ext_c` is the instance bound to the HIP library;
    `allow_cpu` exists only so the test-suite can bind the CPU emulation build of the same sources."""

    def __init__(self, lib: _lib.Lib | None = None, allow_cpu: bool = False):
        self._lib = lib
        self.allow_cpu = allow_cpu

    @property
    def lib(self) -> _lib.Lib:
        if self._lib is None:
            self._lib = _lib.hip_lib()
        return self._lib

    # ---- helpers ----------------------------------------------------------------------------------------------------

    def _ptr(self, t, dtype=None, name="tensor"):
        if t is None or t.is_meta:
            return None
        if dtype is not None and t.dtype is not dtype:
            raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
        if not t.is_contiguous():
            raise RuntimeError(f"{name}: tensor must be contiguous")
        if not t.is_cuda and not (self.allow_cpu and t.device.type == "cpu"):
            raise RuntimeError(f"{name}: tensor must live on a HIP device (got {t.device}); there is no CPU path")
        return t.data_ptr()

    def _stream(self, t) -> int | None:
        if t.is_cuda:
            if _raw_stream is not None:
                return _raw_stream(t.get_device())
            return torch.cuda.current_stream(t.device).cuda_stream
        return None

    def _dev(self, t) -> int:
        return t.device.index if t.device.type == "cuda" else 0

    # ---- q_matrix (ext_qmatrix.cpp) -----------------------------------------------------------------------------------

    def make_q_matrix(self, q_weight, q_perm, q_invperm, q_scale, q_scale_max, q_groups, q_group_map,
                      gptq_qzeros, gptq_scales, gptq_g_idx, bias, temp_dq, max_dq_rows) -> int:
        """ext_qmatrix.cpp:21-111.  Returns an opaque handle (int).  Re-lays `q_weight` out IN PLACE."""
        if q_weight.dtype != torch.int32 or q_weight.dim() != 2:
            raise RuntimeError("make_q_matrix: q_weight must be int32 [rows, width]")
        width = q_weight.shape[1]
        if not _is_none(q_scale):
            groups = q_scale.shape[0]
            if _is_none(q_group_map):
                raise RuntimeError("make_q_matrix: q_group_map required for EXL2 tensors")
            height = q_group_map.shape[0] // 2
            if q_scale.shape[1] * 8 != width:
                raise RuntimeError("make_q_matrix: q_scale and q_weight have incompatible shapes")
            if q_scale_max.shape[0] != groups or q_groups.shape[0] != groups * 2:
                raise RuntimeError("make_q_matrix: q_scale_max / q_groups have incompatible shapes")
        else:
            groups = gptq_qzeros.shape[0]
            height = q_weight.shape[0] * 8
            if gptq_qzeros.shape[1] * 8 != width or tuple(gptq_scales.shape) != (groups, width):
                raise RuntimeError("make_q_matrix: qzeros / scales have incompatible shapes")
        if not _is_none(q_perm) and (q_perm.shape[0] != height or q_invperm.shape[0] != height):
            raise RuntimeError("make_q_matrix: q_perm and q_weight have incompatible shapes")
        g_idx_ptr = None
        g_idx_keep = None
        if not _is_none(gptq_g_idx):
            g_idx_keep = gptq_g_idx.to(device="cpu", dtype=torch.int32).contiguous()
            g_idx_ptr = g_idx_keep.data_ptr()
        handle = C.c_void_p()
        lib = self.lib
        lib.check(lib.exl2_make_q_matrix(
            C.byref(handle), self._dev(q_weight), height, width, groups,
            self._ptr(q_weight, torch.int32, "q_weight"),
            self._ptr(q_perm, torch.int16, "q_perm"), self._ptr(q_invperm, torch.int16, "q_invperm"),
            self._ptr(q_scale, torch.int32, "q_scale"), self._ptr(q_scale_max, torch.float16, "q_scale_max"),
            self._ptr(q_groups, torch.int16, "q_groups"),
            self._ptr(gptq_qzeros, torch.int32, "gptq_qzeros"), self._ptr(gptq_scales, torch.float16, "gptq_scales"),
            g_idx_ptr,
            self._ptr(bias, torch.float16, "bias"), self._ptr(temp_dq, None, "temp_dq"), int(max_dq_rows),
            self._stream(q_weight)))
        return int(handle.value)

    def make_q_matrix_split(self, q_weight, q_perm, q_invperm, q_scale, q_scale_max, q_groups, q_group_map,
                            gptq_qzeros, gptq_scales, gptq_g_idx, bias, temp_dq, max_dq_rows) -> int:
        """ext_qmatrix.cpp:113-187: handle over tensors that were already sliced along the output features for one
        device of a tensor-parallel split (the slicing itself is host code: contiguous `q_weight[:, a:b]`,
        `q_scale[:, a/8:b/8]`, `bias[a:b]`, shared q_perm / q_groups).  EXL2 only, like the reference.  The reference's
        variant merely skips building the group map; here the map is derived from q_groups either way, so this is
        make_q_matrix with the reference's restriction."""
        if _is_none(q_scale) or not _is_none(gptq_qzeros) or not _is_none(gptq_scales) or not _is_none(gptq_g_idx):
            raise RuntimeError("Tensor split not implemented for GPTQ matrices")
        return self.make_q_matrix(q_weight, q_perm, q_invperm, q_scale, q_scale_max, q_groups, q_group_map, gptq_qzeros,
                                  gptq_scales, gptq_g_idx, bias, temp_dq, max_dq_rows)

    def free_q_matrix(self, handle: int) -> None:
        self.lib.check(self.lib.exl2_free_q_matrix(handle))

    def q_matrix_info(self, handle: int) -> dict:
        k, n, g, gq, b = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_longlong()
        self.lib.check(self.lib.exl2_q_matrix_info(handle, C.byref(k), C.byref(n), C.byref(g), C.byref(gq), C.byref(b)))
        return {"height": k.value, "width": n.value, "groups": g.value, "is_gptq": bool(gq.value), "bytes": b.value}

    def reconstruct(self, handle: int, output: torch.Tensor) -> None:
        """ext_qmatrix.cpp:196-210: full fp16 dequant into output [height, width], original row order."""
        info = self.q_matrix_info(handle)
        if output.dtype != torch.float16 or tuple(output.shape) != (info["height"], info["width"]):
            raise RuntimeError("reconstruct: output must be half [height, width]")
        self.lib.check(self.lib.exl2_reconstruct(handle, self._ptr(output, torch.float16, "output"), self._stream(output)))

    def gemm_half_q_half(self, a: torch.Tensor, b: int, c: torch.Tensor, force_cuda: bool = False) -> None:
        """ext_qmatrix.cpp:213-247: c[M, N] = a[M, K] @ W (+ bias).  `force_cuda` (reference: force the quantized
        kernel instead of reconstruct + hgemm) is accepted and ignored: this build always multiplies from the packed
        weights."""
        info = self.q_matrix_info(b)
        if a.dtype != torch.float16 or c.dtype != torch.float16:
            raise RuntimeError("gemm_half_q_half: a and c must be half")
        if a.shape[-1] != info["height"] or c.shape[-1] != info["width"]:
            raise RuntimeError("gemm_half_q_half: a, b and c have incompatible shapes")
        m = a.numel() // info["height"]
        if c.numel() != m * info["width"]:
            raise RuntimeError("gemm_half_q_half: a and c have incompatible shapes")
        if m == 0:
            return                                # zero rows (an expert nobody was routed to): nothing to launch
        self.lib.check(self.lib.exl2_gemm_half_q_half(
            self._ptr(a, torch.float16, "a"), b, self._ptr(c, torch.float16, "c"), m, 1, None, 0, 0, self._stream(a)))

    def make_group_map(self, q_groups: torch.Tensor, num_qrows: int) -> torch.Tensor:
        """ext_qmatrix.cpp:341-361 (CPU): int16 [2K] pairs (group index, rows left in group)."""
        qg = q_groups.to(device="cpu", dtype=torch.int16).contiguous()
        groups = qg.shape[0] // 2
        cap = 2 * 65536
        out = torch.empty((cap,), dtype=torch.int16)
        n = self.lib.check(self.lib.exl2_make_group_map(qg.data_ptr(), groups, int(num_qrows), out.data_ptr(), cap))
        return out[:n].clone()

    # ---- norms / rope / activation (ext_norm.cpp, ext_rope.cpp) --------------------------------------------------------

    def rms_norm(self, x, w, y, epsilon: float) -> None:
        """ext_norm.cpp:22-56: y = rmsnorm(x) * w; x, y fp16 or fp32 [rows, dim]."""
        if w.dtype != torch.float16: raise RuntimeError("rms_norm: w must be half")
        if x.shape[-1] != w.shape[0] or x.shape != y.shape: raise RuntimeError("rms_norm: incompatible shapes")
        dim = x.shape[-1]
        rows = x.numel() // dim
        self.lib.check(self.lib.exl2_rms_norm(self._ptr(x, None, "x"), self._ptr(w, torch.float16, "w"),
                                              self._ptr(y, None, "y"), float(epsilon), rows, dim, 0,
                                              int(x.dtype == torch.float32), int(y.dtype == torch.float32),
                                              self._stream(x)))

    def rms_norm_(self, x, w, epsilon: float) -> None:
        """ext_norm.cpp:58-85 (in place)."""
        self.rms_norm(x, w, x, epsilon)

    def rope_(self, x, sin, cos, past_len: int, num_heads: int, head_dim: int, offsets, neox_style: bool) -> None:
        """ext_rope.cpp:21-62: in-place rotary embedding on x [batch, ..., head_dim]."""
        for t, n in ((x, "x"), (sin, "sin"), (cos, "cos")):
            if t.dtype != torch.float16: raise RuntimeError(f"rope_: {n} must be half")
        if cos.shape[-1] != sin.shape[-1]: raise RuntimeError("sin table does not cos table")
        if not _is_none(offsets) and offsets.dtype != torch.int32: raise RuntimeError("rope_: offsets must be int32")
        batch = x.shape[0]
        rows_per_batch = x.numel() // head_dim // batch
        self.lib.check(self.lib.exl2_rope_qk(self._ptr(x, torch.float16, "x"), None, self._ptr(sin), self._ptr(cos),
                                             batch, rows_per_batch, 0, head_dim, num_heads, 0, int(past_len),
                                             self._ptr(offsets), int(bool(neox_style)), cos.shape[-1], self._stream(x)))

    def act_mul_(self, x, y, act_gelu: bool = False) -> None:
        """x = act(x) * y in place (act_mul_kernel, q_mlp_activation.cuh:54-112)."""
        width = x.shape[-1]
        self.lib.check(self.lib.exl2_act_mul(self._ptr(x, torch.float16, "x"), self._ptr(y, torch.float16, "y"),
                                             x.numel() // width, width, int(act_gelu), None, 0, self._stream(x)))

    # ---- quantized KV cache (ext_cache.cpp:80-269) ---------------------------------------------------------------------

    def _kv_codec(self, fn, k_in, k_out, k_scales, v_in, v_out, v_scales, batch_size, offset, width, page_size,
                  cache_seqlens, block_table, wbits, half_side):
        dim = half_side.shape[2] * half_side.shape[3]
        stride_tokens = half_side.shape[1]
        pages_per_seq = 0
        if page_size:
            batch_size = block_table.shape[0]
            pages_per_seq = block_table.shape[1]
            if cache_seqlens.shape[0] != batch_size: raise RuntimeError("q cache: cache_seqlens / block_table mismatch")
        self.lib.check(fn(self._ptr(k_in), self._ptr(k_out), self._ptr(k_scales), self._ptr(v_in), self._ptr(v_out),
                          self._ptr(v_scales), int(batch_size), dim, stride_tokens, int(offset), int(width),
                          int(page_size), self._ptr(cache_seqlens), self._ptr(block_table), pages_per_seq, int(wbits),
                          self._stream(k_in)))

    def fp16_to_q_kv(self, k_in, k_out, k_scales, v_in, v_out, v_scales, batch_size, offset, width, page_size,
                     cache_seqlens, block_table, wbits) -> None:
        if k_in.dtype != torch.float16 or k_out.dtype != torch.uint8: raise RuntimeError("fp16_to_q_kv: bad dtypes")
        self._kv_codec(self.lib.exl2_fp16_to_q_kv, k_in, k_out, k_scales, v_in, v_out, v_scales, batch_size, offset,
                       width, page_size, cache_seqlens, block_table, wbits, k_in)

    def q_to_fp16_kv(self, k_in, k_out, k_scales, v_in, v_out, v_scales, batch_size, offset, width, page_size,
                     cache_seqlens, block_table, wbits) -> None:
        if k_in.dtype != torch.uint8 or k_out.dtype != torch.float16: raise RuntimeError("q_to_fp16_kv: bad dtypes")
        self._kv_codec(self.lib.exl2_q_to_fp16_kv, k_in, k_out, k_scales, v_in, v_out, v_scales, batch_size, offset,
                       width, page_size, cache_seqlens, block_table, wbits, k_out)

    def _matrix_q4(self, fn, name, half, codes, scales) -> None:
        """ext_qmatrix.cpp:293-335: the Q4 cache codec over a flat array (matrix_*_cuda call array_*_q_kv_cuda with K only,
        one row, width = numel, 4 bits; used for Q4-compressed fp16 head / embedding weights, linear.py:191-212,500)."""
        if half.dtype != torch.float16 or codes.dtype != torch.uint8: raise RuntimeError(f"{name}: bad dtypes")
        if half.numel() != codes.numel() * 2: raise RuntimeError(f"{name}: tensor size mismatch")
        numel = half.numel()
        if numel % 512: raise RuntimeError(f"{name}: numel must be a multiple of 512 (one codec block)")
        if scales.numel() * 32 < numel: raise RuntimeError(f"{name}: scales tensor too small")
        k_in, k_out = (half, codes) if fn is self.lib.exl2_fp16_to_q_kv else (codes, half)
        self.lib.check(fn(self._ptr(k_in), self._ptr(k_out), self._ptr(scales, torch.float16, "scales"), None, None, None,
                          1, 512, numel // 512, 0, numel // 512, 0, None, None, 0, 4, self._stream(half)))

    def matrix_fp16_to_q4(self, in_tensor, out_tensor, scales) -> None:
        self._matrix_q4(self.lib.exl2_fp16_to_q_kv, "matrix_fp16_to_q4", in_tensor, out_tensor, scales)

    def matrix_q4_to_fp16(self, in_tensor, scales, out_tensor) -> None:
        self._matrix_q4(self.lib.exl2_q_to_fp16_kv, "matrix_q4_to_fp16", out_tensor, in_tensor, scales)

    def _fp8(self, fn, name, in_tensor, out_tensor, batch_size, offset, width, in_dtype, out_dtype) -> None:
        if in_tensor.dtype != in_dtype or out_tensor.dtype != out_dtype: raise RuntimeError(f"{name}: bad dtypes")
        if in_tensor.dim() != 4 or tuple(in_tensor.shape) != tuple(out_tensor.shape):
            raise RuntimeError(f"{name}: tensors must be [batch, seq, kv_heads, head_dim] of the same shape")
        if batch_size > in_tensor.shape[0]: raise RuntimeError(f"{name}: batch_size exceeds the cache")
        tsize = in_tensor.shape[2] * in_tensor.shape[3]
        self.lib.check(fn(self._ptr(in_tensor), self._ptr(out_tensor), int(batch_size), in_tensor.shape[1] * tsize, tsize,
                          int(offset), int(width), self._stream(in_tensor)))

    def fp16_to_fp8(self, in_tensor, out_tensor, batch_size: int, offset: int, width: int) -> None:
        """ext_cache.cpp:14-45: FP8 cache store (upper byte of each fp16) of tokens [offset, offset + width)."""
        self._fp8(self.lib.exl2_fp16_to_fp8, "fp16_to_fp8", in_tensor, out_tensor, batch_size, offset, width,
                  torch.float16, torch.uint8)

    def fp8_to_fp16(self, in_tensor, out_tensor, batch_size: int, offset: int, width: int) -> None:
        """ext_cache.cpp:47-78"""
        self._fp8(self.lib.exl2_fp8_to_fp16, "fp8_to_fp16", in_tensor, out_tensor, batch_size, offset, width,
                  torch.uint8, torch.float16)

    def cache_rotate(self, cache, order, temp) -> None:
        """cuda/cache.cu:548-576: cyclic move of cache pages (defragmenter).  cache [num_pages, ...] contiguous, order
        int32 [n] on the cache's device, temp sized as one page (checked like the reference; not used by the kernel)."""
        if cache.dim() < 2: raise RuntimeError("cache argument must have dim >= 2")
        if order.dim() != 1: raise RuntimeError("order argument must have dim == 1")
        if order.dtype != torch.int32: raise RuntimeError("cache_rotate: order must be int32")
        page_bytes = cache.numel() * cache.element_size() // cache.shape[0]
        if temp.numel() * temp.element_size() != page_bytes: raise RuntimeError("temp tensor incorrect size")
        if not cache.is_contiguous(): raise RuntimeError("cache_rotate: cache must be contiguous")
        self.lib.check(self.lib.exl2_cache_rotate(self._ptr(cache), self._ptr(order, torch.int32, "order"), page_bytes,
                                                  order.shape[0], self._stream(cache)))

    def count_match(self, a, b, max_a: int) -> int:
        """ext_cache.cpp:285-302: common-prefix length of two (1, n) int64 CPU tensors, at most min(max_a, b.shape[1])."""
        import ctypes
        if a.dtype != torch.int64 or b.dtype != torch.int64 or a.device.type != "cpu" or b.device.type != "cpu":
            raise RuntimeError("count_match: int64 CPU tensors expected")
        a, b = a.contiguous(), b.contiguous()
        out = ctypes.c_int(0)
        self.lib.check(self.lib.exl2_count_match(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                                                 min(int(max_a), a.shape[-1]), b.shape[-1], ctypes.byref(out)))
        return int(out.value)

    # ---- peer copies of the single-process tensor parallel (csrc/peer.hip; used by ext_tp.py) -------------------------------

    def copy_2d_async(self, dst, src, stream: int | None) -> None:
        """dst[r, :] = src[r, :] for 2-D views of equal shape whose rows are contiguous (dst / src may be column slices of
        wider matrices, on the host (pinned), this device or a peer device); asynchronous on `stream`."""
        if dst.dim() != 2 or src.dim() != 2 or dst.shape != src.shape or dst.dtype != src.dtype:
            raise RuntimeError(f"copy_2d_async: shapes / dtypes differ ({tuple(dst.shape)} {dst.dtype} <- {tuple(src.shape)} {src.dtype})")
        if (dst.shape[1] > 1 and (dst.stride(1) != 1 or src.stride(1) != 1)) or dst.shape[0] == 0 or dst.shape[1] == 0:
            if dst.numel() == 0:
                return
            raise RuntimeError("copy_2d_async: rows must be contiguous")
        es = dst.element_size()
        dp = dst.stride(0) * es if dst.shape[0] > 1 else dst.shape[1] * es
        sp = src.stride(0) * es if src.shape[0] > 1 else src.shape[1] * es
        self.lib.check(self.lib.exl2_memcpy_2d_async(dst.data_ptr(), dp, src.data_ptr(), sp, dst.shape[1] * es, dst.shape[0], stream))

    def release_scratch(self, device=None) -> int:
        """frees the prefill / batched-decode staging buffers of every stream of `device` (default: the current one);
        call only when no captured graph that used them will be replayed (include/exl2_hip.h)"""
        if device is not None and torch.device(device).type == "cuda":
            with torch.cuda.device(device):
                return int(self.lib.exl2_release_scratch(None, 1))
        return int(self.lib.exl2_release_scratch(None, 1))

    def enable_peer_access(self, devices) -> int:
        import ctypes
        devs = [int(d) for d in devices]
        arr = (ctypes.c_int * max(1, len(devs)))(*devs)
        return self.lib.check(self.lib.exl2_enable_peer_access(arr, len(devs)))

    # ---- load path (ext_stloader.cpp) ---------------------------------------------------------------

    def stloader_read(self, filename: str, offset: int, size: int, target) -> None:
        """ext_stloader.cpp:11-157: `size` bytes at `offset` of `filename` -> the contiguous tensor `target` (CPU or device)."""
        if size == 0:
            return
        nbytes = target.numel() * target.element_size()
        if size != nbytes:
            raise RuntimeError(f"stloader_read: {size} bytes requested for a tensor of {nbytes} bytes")
        if not target.is_contiguous():
            raise RuntimeError("stloader_read: target must be contiguous")
        dev = -1 if target.device.type == "cpu" else (target.device.index or 0)
        self.lib.check(self.lib.exl2_stloader_read(os.fsencode(filename), int(offset), int(size), target.data_ptr(), dev,
                                                   self._stream(target)))

    def tensor_remap(self, tensor, index) -> None:
        """ext_stloader.cpp:160-184: in place new[:, c] = old[:, index[c]] (int32 CPU tensors; linear.py:156-158)."""
        if tensor.dtype != torch.int32 or index.dtype != torch.int32 or tensor.dim() != 2 or index.dim() != 1 \
                or index.shape[0] != tensor.shape[1] or tensor.device.type != "cpu" or index.device.type != "cpu":
            raise RuntimeError("tensor_remap: expects CPU int32 [rows, cols] and int32 [cols]")
        if not tensor.is_contiguous():
            raise RuntimeError("tensor_remap: tensor must be contiguous")
        index = index.contiguous()
        self.lib.check(self.lib.exl2_tensor_remap(tensor.data_ptr(), tensor.shape[0], tensor.shape[1], index.data_ptr()))

    def tensor_remap_4bit(self, tensor, index) -> None:
        """ext_stloader.cpp:186-219: the same on 4-bit values packed 8 per int32 along the columns (q_scale)."""
        if tensor.dtype != torch.int32 or index.dtype != torch.int32 or tensor.dim() != 2 or index.dim() != 1 \
                or index.shape[0] != tensor.shape[1] * 8 or tensor.device.type != "cpu" or index.device.type != "cpu":
            raise RuntimeError("tensor_remap_4bit: expects CPU int32 [rows, cols / 8] and int32 [cols]")
        if not tensor.is_contiguous():
            raise RuntimeError("tensor_remap_4bit: tensor must be contiguous")
        index = index.contiguous()
        self.lib.check(self.lib.exl2_tensor_remap_4bit(tensor.data_ptr(), tensor.shape[0], index.shape[0], index.data_ptr()))

    # ---- attention (replaces flash_attn_with_kvcache / _attn_torch) -------------------------------------

    def paged_attn_scratch_bytes(self, rows: int, head_dim: int, nsplit: int) -> int:
        return int(self.lib.exl2_paged_attn_scratch_bytes(rows, head_dim, nsplit))

    def paged_attn(self, q, k_cache, v_cache, out, cache_seqlens, block_table, len_const: int = 0, len_offset: int = 0,
                   softmax_scale: float | None = None, causal: bool = True, nsplit: int = 0, scratch=None,
                   window_left: int = -1, softcap: float = 0.0) -> None:
        """q [b, s, H, hd]; caches [pages, page_size, KVH, hd] (block_table [b, pages]) or [b, T, KVH, hd] (no table).
        window_left / softcap: flash-attn's window_size[0] and softcap (attn.py:590-600); -1 / 0 = off."""
        b, s, nh, hd = q.shape
        kvh = k_cache.shape[2]
        page_size = k_cache.shape[1]
        pps = 0 if _is_none(block_table) else block_table.shape[1]
        scale = hd ** -0.5 if softmax_scale is None else softmax_scale
        sb = 0 if scratch is None else scratch.numel() * scratch.element_size()
        self.lib.check(self.lib.exl2_paged_attn_ex(
            self._ptr(q, torch.float16, "q"), self._ptr(k_cache, torch.float16, "k_cache"),
            self._ptr(v_cache, torch.float16, "v_cache"), self._ptr(out, torch.float16, "out"),
            self._ptr(cache_seqlens, torch.int32, "cache_seqlens"), self._ptr(block_table, torch.int32, "block_table"),
            b, s, nh, kvh, hd, page_size, pps, int(len_const), int(len_offset), float(scale), int(causal), int(nsplit),
            self._ptr(scratch), sb, int(window_left), float(softcap or 0.0), self._stream(q)))

    def flash_prefill(self, q, k_cache, v_cache, out, cache_seqlens, block_table, len_const: int = 0, len_offset: int = 0,
                      softmax_scale: float | None = None, causal: bool = True, window_left: int = -1, softcap: float = 0.0) -> bool:
        """csrc/attn_prefill.hip: paged_attn's contract for many query rows (MFMA flash attention).  False when the head
        size is outside {64, 128, 256} or the pages are shorter than a 64-key tile (nothing launched)."""
        b, s, nh, hd = q.shape
        kvh = k_cache.shape[2]
        page_size = k_cache.shape[1]
        pps = 0 if _is_none(block_table) else block_table.shape[1]
        scale = hd ** -0.5 if softmax_scale is None else softmax_scale
        rc = self.lib.check(self.lib.exl2_flash_prefill_ex(
            self._ptr(q, torch.float16, "q"), self._ptr(k_cache, torch.float16, "k_cache"),
            self._ptr(v_cache, torch.float16, "v_cache"), self._ptr(out, torch.float16, "out"),
            self._ptr(cache_seqlens, torch.int32, "cache_seqlens"), self._ptr(block_table, torch.int32, "block_table"),
            b, s, nh, kvh, hd, page_size, pps, int(len_const), int(len_offset), float(scale), int(causal), int(window_left),
            float(softcap or 0.0), self._stream(q)))
        return rc == 0

    def paged_attn_q4(self, q, k_codes, k_scales, v_codes, v_scales, out, cache_seqlens, block_table, len_const: int = 0,
                      len_offset: int = 0, softmax_scale: float | None = None, causal: bool = True, nsplit: int = 0,
                      scratch=None, k_new=None, v_new=None, out_invperm: int | None = None, counters=None) -> bool:
        """Attention over Q4 codes + scales ([b | pages, T | page_size, KVH, hd/2] uint8, [.., KVH, hd/32] fp16) without
        unpacking them; False when the shape needs the unpack route.  out_invperm (device pointer, u16 [H * hd]): output in
        o_proj's packed order (chained decode).  counters (zeroed int32 tickets, left zeroed): the split partials are merged inside
        the launch instead of by a second one."""
        b, s, nh, hd = q.shape
        kvh = k_codes.shape[2]
        page_size = k_codes.shape[1]
        pps = 0 if _is_none(block_table) else block_table.shape[1]
        scale = hd ** -0.5 if softmax_scale is None else softmax_scale
        sb = 0 if scratch is None else scratch.numel() * scratch.element_size()
        args = (self._ptr(q, torch.float16, "q"), self._ptr(k_codes, torch.uint8, "k_codes"),
                self._ptr(k_scales, torch.float16, "k_scales"), self._ptr(v_codes, torch.uint8, "v_codes"),
                self._ptr(v_scales, torch.float16, "v_scales"), self._ptr(k_new, torch.float16, "k_new"),
                self._ptr(v_new, torch.float16, "v_new"), self._ptr(out, torch.float16, "out"),
                self._ptr(cache_seqlens, torch.int32, "cache_seqlens"), self._ptr(block_table, torch.int32, "block_table"),
                b, s, nh, kvh, hd, page_size, pps, int(len_const), int(len_offset), float(scale), int(causal), int(nsplit),
                self._ptr(scratch), sb, out_invperm or None)
        if counters is not None:
            rc = self.lib.check(self.lib.exl2_paged_attn_q4_merged(*args, self._ptr(counters, torch.int32, "counters"), counters.numel(),
                                                                   self._stream(q)))
        else:
            rc = self.lib.check(self.lib.exl2_paged_attn_q4(*args, self._stream(q)))
        return rc == 0

    def attn_q4_decode_fused(self, q, k_new, v_new, k_codes, k_scales, v_codes, v_scales, out, sin, cos, cache_seqlens, block_table,
                             past_const: int, rope_style: int, scratch, counters, softmax_scale: float | None = None, nsplit: int = 0,
                             out_invperm: int | None = None, sincos_size: int = 0) -> bool:
        """One launch for a decode step over a Q4 cache: RoPE(q, k_new) on the way in (the tensors are NOT modified), Q4 pack of the
        new rows, attention over codes + the step's own fp16 rows, split merge.  False: shape not covered."""
        b, s, nh, hd = q.shape
        kvh = k_new.shape[2]
        page_size = k_codes.shape[1]
        pps = 0 if _is_none(block_table) else block_table.shape[1]
        scale = hd ** -0.5 if softmax_scale is None else softmax_scale
        sb = 0 if scratch is None else scratch.numel() * scratch.element_size()
        rc = self.lib.check(self.lib.exl2_attn_q4_decode_fused(
            self._ptr(q, torch.float16, "q"), self._ptr(k_new, torch.float16, "k_new"), self._ptr(v_new, torch.float16, "v_new"),
            self._ptr(k_codes, torch.uint8, "k_codes"), self._ptr(k_scales, torch.float16, "k_scales"),
            self._ptr(v_codes, torch.uint8, "v_codes"), self._ptr(v_scales, torch.float16, "v_scales"),
            self._ptr(out, torch.float16, "out"), self._ptr(sin), self._ptr(cos),
            self._ptr(cache_seqlens, torch.int32, "cache_seqlens"), self._ptr(block_table, torch.int32, "block_table"),
            b, s, nh, kvh, hd, page_size, pps, int(past_const), float(scale), int(rope_style), int(sincos_size), int(nsplit),
            self._ptr(scratch), sb, self._ptr(counters, torch.int32, "counters"), counters.numel(), out_invperm or None,
            self._stream(q)))
        return rc == 0

    def rope_quant_append_q4(self, q, k_new, v_new, k_codes, k_scales, v_codes, v_scales, sin, cos, past_len: int, past_lens,
                             block_table, rope_style: int, sincos_size: int = 0) -> bool:
        """RoPE(q, k_new) in place + Q4 pack of the rotated k_new and of v_new into codes / scales at past_len (+ past_lens[b]) + j
        through the block table: one launch for rope_kv_append + fp16_to_q_kv of a decode step.  False: shape not covered."""
        b, s, nh, hd = q.shape
        kvh = k_new.shape[2]
        page_size = k_codes.shape[1]
        pps = 0 if _is_none(block_table) else block_table.shape[1]
        rc = self.lib.check(self.lib.exl2_rope_quant_append_q4(
            self._ptr(q, torch.float16, "q"), self._ptr(k_new, torch.float16, "k_new"), self._ptr(v_new, torch.float16, "v_new"),
            self._ptr(k_codes, torch.uint8, "k_codes"), self._ptr(k_scales, torch.float16, "k_scales"),
            self._ptr(v_codes, torch.uint8, "v_codes"), self._ptr(v_scales, torch.float16, "v_scales"),
            self._ptr(sin), self._ptr(cos), b, s, nh, kvh, hd, int(past_len), self._ptr(past_lens), self._ptr(block_table),
            page_size, pps, int(rope_style), int(sincos_size), self._stream(q)))
        return rc == 0

    def rope_kv_append(self, q, k_new, v_new, k_cache, v_cache, sin, cos, past_len: int, past_lens, block_table,
                       rope_style: int, sincos_size: int = 0) -> None:
        b, s, nh, hd = q.shape
        kvh = k_new.shape[2]
        page_size = 0 if _is_none(k_cache) else k_cache.shape[1]
        pps = 0 if _is_none(block_table) else block_table.shape[1]
        self.lib.check(self.lib.exl2_rope_kv_append(
            self._ptr(q, torch.float16, "q"), self._ptr(k_new, torch.float16, "k_new"), self._ptr(v_new),
            self._ptr(k_cache), self._ptr(v_cache), self._ptr(sin), self._ptr(cos), b, s, nh, kvh, hd, int(past_len),
            self._ptr(past_lens), self._ptr(block_table), page_size, pps, int(rope_style), int(sincos_size),
            self._stream(q)))

    def attn_decode_fused(self, q, k_new, v_new, k_cache, v_cache, out, sin, cos, cache_seqlens, block_table,
                          past_const: int, rope_style: int, scratch, counters, softmax_scale: float | None = None,
                          sincos_size: int = 0, nsplit: int = 0, out_invperm: int | None = None) -> bool:
        """One launch for RoPE + append + attention + merge; False when the shape needs the three-launch path.
        out_invperm (device pointer, u16 [H * hd]): write the output in o_proj's packed order (chained decode)."""
        b, s, nh, hd = q.shape
        kvh = k_new.shape[2]
        page_size = k_cache.shape[1]
        pps = 0 if _is_none(block_table) else block_table.shape[1]
        scale = hd ** -0.5 if softmax_scale is None else softmax_scale
        sb = 0 if scratch is None else scratch.numel() * scratch.element_size()
        rc = self.lib.check(self.lib.exl2_attn_decode_fused(
            self._ptr(q, torch.float16, "q"), self._ptr(k_new, torch.float16, "k_new"),
            self._ptr(v_new, torch.float16, "v_new"), self._ptr(k_cache, torch.float16, "k_cache"),
            self._ptr(v_cache, torch.float16, "v_cache"), self._ptr(out, torch.float16, "out"), self._ptr(sin),
            self._ptr(cos), self._ptr(cache_seqlens, torch.int32, "cache_seqlens"),
            self._ptr(block_table, torch.int32, "block_table"), b, s, nh, kvh, hd, page_size, pps, int(past_const),
            float(scale), int(rope_style), int(sincos_size), int(nsplit), self._ptr(scratch), sb,
            self._ptr(counters, torch.int32, "counters"), 0 if counters is None else counters.numel(),
            out_invperm or None, self._stream(q)))
        return rc == 0

    def flash_attn_with_kvcache(self, q, k_cache, v_cache, k=None, v=None, cache_seqlens=None, block_table=None,
                                causal: bool = True, softmax_scale: float | None = None, scratch=None,
                                window_left: int = -1, softcap: float = 0.0):
        """Drop-in for flash_attn.flash_attn_with_kvcache as the reference calls it (attn.py:602-613; window_size[0] / softcap: :590-600)."""
        out = torch.empty_like(q)
        s = q.shape[1]
        off = s if k is not None else 0
        if k is not None:
            self.rope_kv_append(q, k, v, k_cache, v_cache, none_tensor, none_tensor, 0, cache_seqlens, block_table, 0)
        # many query rows (prefill chunks): MFMA flash attention; decode-shaped: the split-KV kernel
        if not (s > 16 and self.flash_prefill(q, k_cache, v_cache, out, cache_seqlens, block_table, 0, off, softmax_scale, causal, window_left, softcap)):
            self.paged_attn(q, k_cache, v_cache, out, cache_seqlens, block_table, 0, off, softmax_scale, causal, 0, scratch, window_left, softcap)
        return out

    # ---- fused modules (ext_qattn.cpp, ext_qmlp.cpp) -------------------------------------------------------------------

    def make_q_attn(self, layernorm, layernorm_bias, layernorm_is_rms, headnorm_is_rms, norm_epsilon, q_q_proj,
                    q_k_proj, q_v_proj, q_o_proj, temp_state, temp_dq, max_rows, hidden_size, num_heads, num_kv_heads,
                    head_dim, max_seq_len, has_residual, rope_style, sincos_size, q_norm, k_norm, post_layernorm,
                    post_layernorm_bias, residual_fp32, use_graphs) -> int:
        h = C.c_void_p()
        self.lib.check(self.lib.exl2_make_q_attn(
            C.byref(h), self._ptr(layernorm), self._ptr(layernorm_bias), int(layernorm_is_rms), int(headnorm_is_rms),
            float(norm_epsilon), q_q_proj, q_k_proj, q_v_proj, q_o_proj, self._ptr(temp_state), self._ptr(temp_dq),
            int(max_rows), int(hidden_size), int(num_heads), int(num_kv_heads), int(head_dim), int(max_seq_len),
            int(has_residual), int(rope_style), int(sincos_size), self._ptr(q_norm), self._ptr(k_norm),
            self._ptr(post_layernorm), self._ptr(post_layernorm_bias), int(residual_fp32), int(use_graphs)))
        return int(h.value)

    def free_q_attn(self, handle: int) -> None:
        self.lib.check(self.lib.exl2_free_q_attn(handle))

    def q_attn_forward_1(self, q_attn, x, batch_size, q_len, past_len, past_lens, q_temp, k_temp, v_temp, sin, cos,
                         loras=None, loras_temp=None, apply_rope: bool = True) -> None:
        if loras: raise RuntimeError("q_attn_forward_1: LoRA is out of scope of this build")
        self.lib.check(self.lib.exl2_q_attn_forward_1(
            q_attn, self._ptr(x, torch.float16, "x"), int(batch_size), int(q_len), int(past_len),
            self._ptr(past_lens, torch.int32, "past_lens"), self._ptr(q_temp, torch.float16, "q_temp"),
            self._ptr(k_temp, torch.float16, "k_temp"), self._ptr(v_temp, torch.float16, "v_temp"),
            self._ptr(sin), self._ptr(cos), int(apply_rope), self._stream(x)))

    def q_attn_forward_2(self, q_attn, x, attn_output, batch_size, q_len, loras=None, loras_temp=None) -> None:
        if loras: raise RuntimeError("q_attn_forward_2: LoRA is out of scope of this build")
        self.lib.check(self.lib.exl2_q_attn_forward_2(q_attn, self._ptr(x, torch.float16, "x"),
                                                      self._ptr(attn_output, torch.float16, "attn_output"),
                                                      int(batch_size), int(q_len), self._stream(x)))

    def make_q_mlp(self, layernorm, layernorm_bias, layernorm_is_rms, norm_epsilon, q_gate, q_up, q_down, temp_state,
                   temp_a, temp_b, temp_dq, max_rows, act_gelu, has_residual, post_layernorm, post_layernorm_bias,
                   residual_fp32, use_graphs) -> int:
        h = C.c_void_p()
        self.lib.check(self.lib.exl2_make_q_mlp(
            C.byref(h), self._ptr(layernorm), self._ptr(layernorm_bias), int(layernorm_is_rms), float(norm_epsilon),
            q_gate or None, q_up, q_down, self._ptr(temp_state), self._ptr(temp_a), self._ptr(temp_b),
            self._ptr(temp_dq), int(max_rows), int(act_gelu), int(has_residual), self._ptr(post_layernorm),
            self._ptr(post_layernorm_bias), int(residual_fp32), int(use_graphs)))
        return int(h.value)

    def free_q_mlp(self, handle: int) -> None:
        self.lib.check(self.lib.exl2_free_q_mlp(handle))

    def q_mlp_forward_(self, q_mlp, x, loras=None, loras_temp=None) -> None:
        if loras: raise RuntimeError("q_mlp_forward_: LoRA is out of scope of this build")
        hidden = x.shape[-1]
        self.lib.check(self.lib.exl2_q_mlp_forward(q_mlp, self._ptr(x, torch.float16, "x"), x.numel() // hidden,
                                                   self._stream(x)))

    # ---- MoE (ext_qmlp.cpp:245-272) -------------------------------------------------------------------------------------

    def make_q_moe_mlp(self, layernorm, layernorm_bias, layernorm_is_rms, norm_epsilon, gate, num_experts,
                       num_experts_per_token, w1, w2, w3, temp_state, temp_gathered_state, temp_a, temp_b, temp_logits,
                       temp_dq, max_rows, act_gelu) -> int:
        if not (len(w1) == len(w2) == len(w3) == num_experts):
            raise RuntimeError("make_q_moe_mlp: expert handle lists must have num_experts entries")
        arr = lambda hs: (C.c_void_p * num_experts)(*[C.c_void_p(int(h)) for h in hs])
        h = C.c_void_p()
        self.lib.check(self.lib.exl2_make_q_moe_mlp(
            C.byref(h), self._ptr(layernorm), self._ptr(layernorm_bias), int(layernorm_is_rms), float(norm_epsilon),
            self._ptr(gate, torch.float16, "gate"), int(num_experts), int(num_experts_per_token), arr(w1), arr(w2), arr(w3),
            self._ptr(temp_state), self._ptr(temp_gathered_state), self._ptr(temp_a), self._ptr(temp_b),
            self._ptr(temp_logits), self._ptr(temp_dq), int(max_rows), int(act_gelu)))
        return int(h.value)

    def free_q_moe_mlp(self, handle: int) -> None:
        self.lib.check(self.lib.exl2_free_q_moe_mlp(handle))

    def q_moe_mlp_forward_(self, q_moe_mlp, x) -> None:
        """In place on x [rows, hidden] (any row count up to max_rows; the reference's fused path stops at 4)."""
        hidden = x.shape[-1]
        self.lib.check(self.lib.exl2_q_moe_mlp_forward(q_moe_mlp, self._ptr(x, torch.float16, "x"), x.numel() // hidden,
                                                       self._stream(x)))

    def moe_route(self, x, gate, logits, topk: int) -> None:
        rows, hidden = x.shape
        self.lib.check(self.lib.exl2_moe_route(self._ptr(x, torch.float16, "x"), self._ptr(gate, torch.float16, "gate"),
                                               self._ptr(logits, torch.float16, "logits"), rows, hidden, gate.shape[0],
                                               int(topk), self._stream(x)))

    # ---- chained decode (csrc/qgemv_flat.hip; ours) -----------------------------------------------------------------------

    def q_attn_chain_info(self, q_attn: int):
        """(capable, in_invperm pointer or None, o_invperm pointer or None, pointer to the norm weight in q/k/v's packed order)"""
        cap, a, b, w = C.c_int(0), C.c_void_p(), C.c_void_p(), C.c_void_p()
        self.lib.check(self.lib.exl2_q_attn_chain_info(q_attn, C.byref(cap), C.byref(a), C.byref(b), C.byref(w)))
        return bool(cap.value), a.value, b.value, w.value

    def q_mlp_chain_info(self, q_mlp: int):
        """(capable, in_invperm pointer or None, pointer to the norm weight in gate/up's packed order)"""
        cap, a, w = C.c_int(0), C.c_void_p(), C.c_void_p()
        self.lib.check(self.lib.exl2_q_mlp_chain_info(q_mlp, C.byref(cap), C.byref(a), C.byref(w)))
        return bool(cap.value), a.value, w.value

    def _w_ptr(self, w):
        """next_norm_w: a raw device pointer (from *_chain_info), an fp16 tensor, or None (= all ones)"""
        if w is None or isinstance(w, int):
            return w or None
        return self._ptr(w, torch.float16, "next_norm_w")

    def q_matrix_perm_info(self, q_handle: int):
        """(q_perm pointer or None, q_invperm pointer or None) of a q_matrix (u16 device arrays)"""
        a, b = C.c_void_p(), C.c_void_p()
        self.lib.check(self.lib.exl2_q_matrix_perm_info(q_handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def q_attn_forward_1_chain(self, q_attn, xp, ss, npart: int, rows: int, q_temp, k_temp, v_temp) -> None:
        self.lib.check(self.lib.exl2_q_attn_forward_1_chain(
            q_attn, self._ptr(xp, torch.float16, "xp"), self._ptr(ss, torch.float32, "ss"), int(npart), int(rows),
            self._ptr(q_temp, torch.float16, "q_temp"), self._ptr(k_temp, torch.float16, "k_temp"),
            self._ptr(v_temp, torch.float16, "v_temp"), self._stream(xp)))

    def q_attn_forward_2_chain(self, q_attn, x, attn_out_packed, rows: int, next_invperm, next_norm_w, xp_out, ss_out) -> int:
        """x += attn_out . Wo; publishes xp_out = x * next_norm_w in the next consumer's order + partial sums of x^2.
        Returns the number of partial sums per row written to ss_out."""
        n = C.c_int(0)
        self.lib.check(self.lib.exl2_q_attn_forward_2_chain(
            q_attn, self._ptr(x, torch.float16, "x"), self._ptr(attn_out_packed, torch.float16, "attn_out"), int(rows),
            next_invperm or None, self._w_ptr(next_norm_w), self._ptr(xp_out, torch.float16, "xp_out"),
            self._ptr(ss_out, torch.float32, "ss_out"), C.byref(n), self._stream(x)))
        return n.value

    def q_mlp_forward_chain(self, q_mlp, x, xp, ss, npart: int, rows: int, next_invperm, next_norm_w, xp_out, ss_out) -> int:
        n = C.c_int(0)
        self.lib.check(self.lib.exl2_q_mlp_forward_chain(
            q_mlp, self._ptr(x, torch.float16, "x"), self._ptr(xp, torch.float16, "xp"), self._ptr(ss, torch.float32, "ss"),
            int(npart), int(rows), next_invperm or None, self._w_ptr(next_norm_w), self._ptr(xp_out, torch.float16, "xp_out"),
            self._ptr(ss_out, torch.float32, "ss_out"), C.byref(n), self._stream(x)))
        return n.value

    def q_mlp_forward_chain_part(self, q_mlp, part: int, row0: int, x, xp, ss, npart: int, rows: int, next_invperm, next_norm_w,
                                 xp_out, ss_out) -> int:
        """one half of q_mlp_forward_chain for rows [row0, row0 + rows) (tensors are the row-group slices): part 1 = gate | up,
        part 2 = down (returns the partial sums per row it published)"""
        n = C.c_int(0)
        self.lib.check(self.lib.exl2_q_mlp_forward_chain_part(
            q_mlp, int(part), int(row0), self._ptr(x, torch.float16, "x"), self._ptr(xp, torch.float16, "xp"),
            self._ptr(ss, torch.float32, "ss"), int(npart), int(rows), next_invperm or None, self._w_ptr(next_norm_w),
            self._ptr(xp_out, torch.float16, "xp_out"), self._ptr(ss_out, torch.float32, "ss_out"), C.byref(n), self._stream(x)))
        return n.value

    def q_moe_mlp_forward_chain(self, q_moe, x, rows: int, next_invperm, next_norm_w, xp_out, ss_out) -> int:
        """q_moe_mlp_forward_ in place on x + the hand-off for the next consumer; returns the partial sums per row it published"""
        n = C.c_int(0)
        self.lib.check(self.lib.exl2_q_moe_mlp_forward_chain(
            q_moe, self._ptr(x, torch.float16, "x"), int(rows), next_invperm or None, self._w_ptr(next_norm_w),
            self._ptr(xp_out, torch.float16, "xp_out"), self._ptr(ss_out, torch.float32, "ss_out"), C.byref(n), self._stream(x)))
        return n.value

    def publish_rows(self, x, rows: int, hidden: int, next_invperm, next_norm_w, xp_out, ss_out) -> None:
        """the hand-off a chained producer would have left, made from rows already in memory: xp_out = x * next_norm_w in the
        consumer's packed order, ss_out[row] = sum of squares (one partial per row)"""
        self.lib.check(self.lib.exl2_publish_rows(
            self._ptr(x, torch.float16, "x"), int(rows), int(hidden), next_invperm or None, self._w_ptr(next_norm_w),
            self._ptr(xp_out, torch.float16, "xp_out"), self._ptr(ss_out, torch.float32, "ss_out"), self._stream(x)))

    def gemm_half_q_half_chain(self, xp, ss, npart: int, eps: float, q_handle: int, c, rows: int) -> None:
        """c = rmsnorm(x) . W from xp = x * norm weight (applied by xp's producer, in W's packed order) and ss"""
        self.lib.check(self.lib.exl2_gemm_half_q_half_chain(
            self._ptr(xp, torch.float16, "xp"), self._ptr(ss, torch.float32, "ss"), int(npart),
            float(eps), q_handle, self._ptr(c, torch.float16, "c"), int(rows), self._stream(xp)))

    def chain_route_counts(self, reset: bool = False):
        """(launches taken by csrc/qgemv_lean.hip, launches taken by csrc/qgemv_flat.hip) since the last reset"""
        a, b = C.c_longlong(0), C.c_longlong(0)
        self.lib.check(self.lib.exl2_chain_route_counts(C.byref(a), C.byref(b), 1 if reset else 0))
        return a.value, b.value

    def prefill_route_info(self):
        """(rows, tile rows, weights pre-decoded by wfrag_kernel, calls so far) of the last prefill q_gemm call (>= 129 rows)"""
        out = (C.c_int * 4)()
        self.lib.check(self.lib.exl2_prefill_route_info(out))
        return out[0], out[1] * 32, bool(out[2]), out[3]

    SYNC_BLOCK_WORDS = 1024

    def chain_overlap_begin(self, flags, stream_a, stream_b) -> None:
        """csrc/chain_sync.h: until chain_overlap_end() the chained launches alternate between the two streams and carry
        their dependency in `flags` (int32 [n, SYNC_BLOCK_WORDS], one block per launch, zero at first use)."""
        n = flags.numel() // self.SYNC_BLOCK_WORDS
        self.lib.check(self.lib.exl2_chain_overlap_begin(self._ptr(flags, torch.int32, "flags"), n, stream_a, stream_b))

    def chain_overlap_end(self) -> int:
        n = C.c_int(0)
        self.lib.check(self.lib.exl2_chain_overlap_end(C.byref(n)))
        return n.value

    def chain_set_tiled(self, on: bool) -> None:
        """5..16 rows: the chain's xp buffers in the MFMA-tiled layout between (True) and (False) (include/exl2_hip.h)"""
        self.lib.check(self.lib.exl2_chain_set_tiled(1 if on else 0))

    def embed_rows_chain(self, table, ids, x, next_invperm, next_norm_w, xp_out, ss_out) -> None:
        self.lib.check(self.lib.exl2_embed_rows_chain(
            self._ptr(table, torch.float16, "table"), self._ptr(ids, torch.int32, "ids"), self._ptr(x, torch.float16, "x"),
            ids.numel(), table.shape[1], table.shape[0], next_invperm or None, self._w_ptr(next_norm_w),
            self._ptr(xp_out, torch.float16, "xp_out"), self._ptr(ss_out, torch.float32, "ss_out"), self._stream(x)))

    def gather_f16(self, src, perm_ptr, dst) -> None:
        self.lib.check(self.lib.exl2_gather_f16(self._ptr(src, torch.float16, "src"), perm_ptr or None,
                                                self._ptr(dst, torch.float16, "dst"), src.numel(), self._stream(src)))

    # ---- decode-loop utilities + graphs (ours; no reference counterpart at the ext_c level) ----------------------------

    def embed_rows(self, table, ids, out) -> None:
        self.lib.check(self.lib.exl2_embed_rows(self._ptr(table, torch.float16, "table"),
                                                self._ptr(ids, torch.int32, "ids"), self._ptr(out, torch.float16, "out"),
                                                ids.numel(), table.shape[1], table.shape[0], self._stream(out)))

    def argmax_rows(self, logits, out_ids, vocab: int | None = None, history=None, hist_pos=None, pos_inc: int = 0) -> None:
        """Greedy sampling on the device; optionally logs the token at history[row, hist_pos[row] + pos_inc] and, when
        pos_inc != 0, advances hist_pos by it (the decode loop's position increment, in the same launch)."""
        ld = logits.shape[-1]
        self.lib.check(self.lib.exl2_argmax_rows(self._ptr(logits, torch.float16, "logits"),
                                                 self._ptr(out_ids, torch.int32, "out_ids"), logits.numel() // ld,
                                                 int(vocab or ld), ld, self._ptr(history, torch.int32, "history"),
                                                 self._ptr(hist_pos, torch.int32, "hist_pos"),
                                                 0 if history is None else history.shape[-1], int(pos_inc), self._stream(logits)))

    def sample_rows(self, logits, temperature: float, top_k: int, top_p: float, min_p: float, random: float,
                    out_tokens, out_probs, logit_filter=None, workspace=None, vocab: int | None = None):
        """Temperature / top-k / top-p / min-p sampling on the device, one token per row, with the reference's candidate
        order, thresholds and random recurrence (sample_basic, ext_sampling.cpp:93-301; csrc/sampling.hip).  logits fp16 or
        fp32 [..., ld]; out_tokens int32 [rows], out_probs fp32 [rows]; logit_filter bool / uint8 [rows, vocab] or None;
        workspace fp32 [rows, vocab] (allocated here when None).  Returns the workspace (the rows' probabilities)."""
        ld = logits.shape[-1]
        rows = logits.numel() // ld
        v = int(vocab or ld)
        if logits.dtype not in (torch.float16, torch.float32):
            raise RuntimeError(f"sample_rows: logits must be fp16 or fp32, got {logits.dtype}")
        if workspace is None:
            workspace = torch.empty((rows, v), dtype=torch.float32, device=logits.device)
        if workspace.numel() < rows * v:
            raise RuntimeError("sample_rows: workspace smaller than rows x vocab")
        f = None
        if not _is_none(logit_filter):
            f = logit_filter.view(torch.uint8) if logit_filter.dtype == torch.bool else logit_filter
            if f.numel() != rows * v:
                raise RuntimeError("sample_rows: logit_filter must be [rows, vocab]")
        self.lib.check(self.lib.exl2_sample_rows(self._ptr(logits, None, "logits"), int(logits.dtype == torch.float32), rows, v, ld,
                                                 self._ptr(f, torch.uint8, "logit_filter"), float(temperature), int(top_k),
                                                 float(top_p), float(min_p), float(random),
                                                 self._ptr(out_tokens, torch.int32, "out_tokens"),
                                                 self._ptr(out_probs, torch.float32, "out_probs"),
                                                 self._ptr(workspace, torch.float32, "workspace"), self._stream(logits)))
        return workspace

    def sample_rows_step(self, logits, temperature: float, top_k: int, top_p: float, min_p: float, randoms, counter,
                         out_tokens, out_probs, workspace, history, hist_pos, pos_inc: int = 1, vocab: int | None = None) -> None:
        """sample_rows as the last launch of a decode-step graph (include/exl2_hip.h exl2_sample_rows_step): the random point is
        randoms[counter[0] % len(randoms)] (fp32 / int32 device tensors), the token is logged and the position advanced like
        argmax_rows does.  The caller appends add_i32_(counter, 1)."""
        ld = logits.shape[-1]
        rows = logits.numel() // ld
        v = int(vocab or ld)
        if workspace.numel() < rows * v:
            raise RuntimeError("sample_rows_step: workspace smaller than rows x vocab")
        self.lib.check(self.lib.exl2_sample_rows_step(
            self._ptr(logits, None, "logits"), int(logits.dtype == torch.float32), rows, v, ld, None,
            float(temperature), int(top_k), float(top_p), float(min_p),
            self._ptr(randoms, torch.float32, "randoms"), randoms.numel(), self._ptr(counter, torch.int32, "counter"),
            self._ptr(out_tokens, torch.int32, "out_tokens"), self._ptr(out_probs, torch.float32, "out_probs"),
            self._ptr(workspace, torch.float32, "workspace"), self._ptr(history, torch.int32, "history"),
            self._ptr(hist_pos, torch.int32, "hist_pos"), history.shape[-1], int(pos_inc), self._stream(logits)))

    def add_i32_(self, t, value: int) -> None:
        self.lib.check(self.lib.exl2_add_i32(self._ptr(t, torch.int32, "t"), t.numel(), int(value), self._stream(t)))

    def graph_begin_capture(self, stream: int | None) -> None:
        self.lib.check(self.lib.exl2_graph_begin_capture(stream))

    def graph_end_capture(self, stream: int | None) -> int:
        h = C.c_void_p()
        self.lib.check(self.lib.exl2_graph_end_capture(stream, C.byref(h)))
        return int(h.value)

    def graph_launch(self, graph: int, stream: int | None) -> None:
        self.lib.check(self.lib.exl2_graph_launch(graph, stream))

    def graph_free(self, graph: int) -> None:
        self.lib.check(self.lib.exl2_graph_free(graph))

    # ---- python-level adapter (exllamav2/ext.py:325-410) --------------------------------------------------------------

    def make_q_matrix_from_dict(self, w: dict, temp_dq, key: str | None = None, prescale: float = 1,
                                max_dq_rows: int = 0, offset_qzeros: bool = False, split: bool = False) -> int:
        """The reference's `ext.make_q_matrix(w, temp_dq, ...)` adapter: EXL2 (:334-357) or GPTQ (:361-410) tensors."""
        if "q_weight" in w:
            w["q_scale_max"] *= prescale / 256
            if "q_perm" in w: w["q_perm"] = w["q_perm"].short()
            if "q_invperm" in w: w["q_invperm"] = w["q_invperm"].short()
            if "q_group_map" not in w:
                w["q_group_map"] = self.make_group_map(w["q_groups"], w["q_weight"].shape[0]).to(w["q_groups"].device)
            mk = self.make_q_matrix_split if split else self.make_q_matrix
            return mk(w["q_weight"], w.get("q_perm", none_tensor), w.get("q_invperm", none_tensor),
                      w["q_scale"], w["q_scale_max"], w["q_groups"], w["q_group_map"],
                      none_tensor, none_tensor, none_tensor, w.get("bias", none_tensor), temp_dq, max_dq_rows)
        elif "qweight" in w:
            if prescale != 1: w["scales"] *= prescale
            if w["scales"].dtype == torch.float: w["scales"] = w["scales"].half()
            if offset_qzeros:
                w["qzeros"] -= 0b00010001000100010001000100010001
            if "g_idx" in w and not (w["g_idx"] == 0).all().item():
                w["q_perm"] = torch.empty((w["qweight"].shape[0] * 8,), dtype=torch.short, device=w["qweight"].device)
                w["q_invperm"] = torch.empty_like(w["q_perm"])
                return self.make_q_matrix(w["qweight"], w["q_perm"], w["q_invperm"], none_tensor, none_tensor,
                                          none_tensor, none_tensor, w["qzeros"], w["scales"], w["g_idx"].cpu(),
                                          w.get("bias", none_tensor), temp_dq, max_dq_rows)
            return self.make_q_matrix(w["qweight"], none_tensor, none_tensor, none_tensor, none_tensor, none_tensor,
                                      none_tensor, w["qzeros"], w["scales"], none_tensor,
                                      w.get("bias", none_tensor), temp_dq, max_dq_rows)
        raise RuntimeError("make_q_matrix: neither EXL2 nor GPTQ tensors in dict")


def _device_scoped(fn):
    """The reference opens `const at::cuda::OptionalCUDAGuard device_guard(device_of(x))` at the top of every binding
    (e.g. ext_qmatrix.cpp:41, ext_qattn.cpp:131): the op runs on the device of its tensors, whatever the caller's current
    device is, and the caller's device is put back afterwards.  Same here: the first HIP tensor among the arguments
    names the device; the switch only happens when it differs from the current one (gpu_split / one process, many GPUs).
    Handle-only calls (free_*, *_info) are scoped inside the library by the device stored in the handle."""
    @functools.wraps(fn)
    def scoped(self, *args, **kwargs):
        for a in args:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if a.get_device() != (_cur_device() if _cur_device is not None else torch.cuda.current_device()):
                    with torch.cuda.device(a.device):
                        return fn(self, *args, **kwargs)
                break
        return fn(self, *args, **kwargs)
    return scoped


for _name, _fn in list(vars(ExtC).items()):
    if not _name.startswith("_") and callable(_fn) and not isinstance(_fn, (property, staticmethod, classmethod)):
        setattr(ExtC, _name, _device_scoped(_fn))
del _name, _fn


ext_c = ExtC()
