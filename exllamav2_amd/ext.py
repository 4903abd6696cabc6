"""Host-side mirror of the reference's operator boundary `ext_c.*` (exllamav2/ext.py:291, the pybind module
`exllamav2_ext`, ext_bindings.cpp:27-138) on top of the C-ABI HIP library.

Same names, argument order and meaning as the reference bindings; tensors are torch tensors, handles are python ints,
"None" tensors are passed as the meta-device sentinel `none_tensor` (ext.py:296).  Argument errors raise RuntimeError
like TORCH_CHECK does.  PyTorch is plumbing here: it owns device memory and streams; all arithmetic on this path runs
in the hand-written gfx950 kernels.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import Exl2Error

none_tensor = torch.empty((1, 1), device="meta")


def _is_none(t) -> bool:
    return t is None or (isinstance(t, torch.Tensor) and t.device.type == "meta")


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
        if _is_none(t):
            return None
        if dtype is not None and t.dtype != dtype:
            raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
        if not t.is_contiguous():
            raise RuntimeError(f"{name}: tensor must be contiguous")
        if t.device.type != "cuda" and not (self.allow_cpu and t.device.type == "cpu"):
            raise RuntimeError(f"{name}: tensor must live on a HIP device (got {t.device}); there is no CPU path")
        return t.data_ptr()

    def _stream(self, t) -> int | None:
        if t.device.type == "cuda":
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

    # ---- python-level adapter (exllamav2/ext.py:325-410) --------------------------------------------------------------

    def make_q_matrix_from_dict(self, w: dict, temp_dq, key: str | None = None, prescale: float = 1,
                                max_dq_rows: int = 0, offset_qzeros: bool = False) -> int:
        """The reference's `ext.make_q_matrix(w, temp_dq, ...)` adapter: EXL2 (:334-357) or GPTQ (:361-410) tensors."""
        if "q_weight" in w:
            w["q_scale_max"] *= prescale / 256
            if "q_perm" in w: w["q_perm"] = w["q_perm"].short()
            if "q_invperm" in w: w["q_invperm"] = w["q_invperm"].short()
            if "q_group_map" not in w:
                w["q_group_map"] = self.make_group_map(w["q_groups"], w["q_weight"].shape[0]).to(w["q_groups"].device)
            return self.make_q_matrix(w["q_weight"], w.get("q_perm", none_tensor), w.get("q_invperm", none_tensor),
                                      w["q_scale"], w["q_scale_max"], w["q_groups"], w["q_group_map"],
                                      none_tensor, none_tensor, none_tensor, w.get("bias", none_tensor),
                                      temp_dq, max_dq_rows)
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


ext_c = ExtC()
