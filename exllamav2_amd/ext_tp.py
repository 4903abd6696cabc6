"""Host-side mirror of the reference's single-process tensor-parallel bindings (exllamav2_ext/ext_tp.cpp:26-495,
ext_qattn.cpp:261-700 `tp_attn_forward_*`, ext_qmlp.cpp:326-473 `tp_mlp_forward_`, ext_qmatrix.cpp:249-289
`gemm_half_q_half_tp`, ext_norm.cpp:59-93 `rms_norm_tp`).

What the reference does there is HOST code: one process drives all devices, loops over them inside each binding, stages
every exchange through two pinned host buffers (`cudaMemcpyAsync` D2H, then H2D on each device's stream) and orders the
streams with events.  The arithmetic in between is the per-device kernels of the quantized forward path.  Same here: the
staging and ordering are torch stream / event / copy calls (plumbing), the arithmetic is libexl2_hip.so through
`exllamav2_amd.ext_c` -- rms_norm, q_gemm on the per-device column slices, RoPE, the KV-cache attention, act_mul.

Semantics kept from the reference, because its Python host code relies on them:
  * results of a gather land in `pinned_temp[buffer]` (host memory): the next module reads its input from there
    (tensor_p.py:252-256, 296-298), and a CPU `source` of `tp_broadcast` is taken to BE that buffer (ext_tp.cpp:146-156);
  * everything is asynchronous on the per-device streams handed to `make_tp_context` (tensor_p.py:196-199); the caller
    synchronises (model.py:1031-1032 `wait_streams`);
  * `t_device = -1` only: `TP_MULTITHREADED` is compiled out in the reference (ext_tp.h:10), so is the per-thread form here.

The multi-process (one rank per GPU, RCCL) tensor parallel of this package is exllamav2_amd/tensor_p.py; this module exists
so that the UNMODIFIED reference (`model.load_tp`, `ExLlamaV2Cache_TP`) runs on the drop-in.
"""
from __future__ import annotations

import contextlib

import torch

from . import ext as _ext_mod
from .ext import none_tensor


class _Lib:
    """the operator surface in use: `exllamav2_amd.ext_c` (tests bind the CPU emulation build instead)"""
    bound = None

    def __getattr__(self, name):
        return getattr(_Lib.bound or _ext_mod.ext_c, name)


_e = _Lib()

BROADCAST_KV, BROADCAST_ID, BROADCAST_VC, BROADCAST_RS, BROADCAST_Q = 0, 1, 2, 3, 4          # ext_tp.h:15-19
ROPE_STYLE_NONE, ROPE_STYLE_GPTJ, ROPE_STYLE_NEOX = 0, 1, 2                                      # q_attn.cuh:13-15


class ExtTPContext:
    """ext_tp.h:60-105 / ext_tp.cpp:26-80.  Splits are lists of (device, first, last) in units of the split's quantity."""

    def __init__(self, kv_split, id_split, vc_split, rs_split, q_split, pinned_temp, streams):
        self.splits = [list(map(tuple, s)) for s in (kv_split, id_split, vc_split, rs_split, q_split)]
        self.pinned_temp = list(pinned_temp)
        for pt in self.pinned_temp:
            if pt.device.type != "cpu" or pt.dtype != torch.float16:
                raise RuntimeError("make_tp_context: pinned_temp must be half tensors in host memory")
        self.pinned_size = self.pinned_temp[-1].numel() * 2 if self.pinned_temp else 0
        self.stream_handles = [int(s) for s in streams]
        self.all_devices = [i for i, s in enumerate(self.stream_handles) if s]
        # host-only mode: the test-suite drives this module with CPU tensors on the emulation build of the library (every
        # "device" of the split is the host; copies and kernels complete in program order, streams / events are moot)
        self.host_only = not torch.cuda.is_available()
        if not self.host_only:
            self.streams = {d: torch.cuda.ExternalStream(self.stream_handles[d], device=d) for d in self.all_devices}
            self.sync_events = {d: torch.cuda.Event() for d in self.all_devices}

    def split(self, broadcast_type: int):
        return self.splits[broadcast_type]

    def on(self, dev):
        """device + stream scope of one device (the reference's cudaSetDevice + ctx->streams[dev])"""
        if self.host_only:
            return contextlib.nullcontext()
        return torch.cuda.stream(self.streams[dev])

    def record(self, dev):
        if not self.host_only:
            self.sync_events[dev].record(self.streams[dev])

    def wait(self, dev, other):
        if not self.host_only:
            self.streams[dev].wait_event(self.sync_events[other])


_contexts: dict[int, ExtTPContext] = {}


def make_tp_context(kv_split, id_split, vc_split, rs_split, q_split, pinned_temp, streams) -> int:
    ctx = ExtTPContext(kv_split, id_split, vc_split, rs_split, q_split, pinned_temp, streams)
    if not ctx.host_only and len(ctx.all_devices) > 1:
        ctx.unreachable_pairs = _e.enable_peer_access(ctx.all_devices)          # xGMI peer copies in tp_gather / tp_broadcast
    _contexts[id(ctx)] = ctx
    return id(ctx)


def free_tp_context(tp_context: int) -> None:
    _contexts.pop(tp_context, None)


def _ctx(tp_context: int) -> ExtTPContext:
    try:
        return _contexts[tp_context]
    except KeyError:
        raise RuntimeError("invalid tensor-parallel context handle") from None


def _pinned_view(ctx: ExtTPContext, buffer: int, rows: int, cols: int, dtype=torch.float16):
    esize = torch.empty((), dtype=dtype).element_size()
    if rows * cols * esize > ctx.pinned_size:
        raise RuntimeError("Temporary tensor is too small")                           # ext_tp.cpp:143
    flat = ctx.pinned_temp[buffer].view(torch.uint8)[:rows * cols * esize]
    return flat.view(dtype).view(rows, cols)


def tp_cross_device_barrier(tp_context: int, broadcast_type: int = 0, t_device: int = -1, stage: int = -1,
                            next_stage: int = -1) -> None:
    """ext_tp.cpp:289-355: every device's stream waits for what every other device's stream has been given so far."""
    ctx = _ctx(tp_context)
    for d in ctx.all_devices:
        ctx.record(d)
    for d in ctx.all_devices:
        for o in ctx.all_devices:
            if o != d:
                ctx.wait(d, o)


def tp_broadcast(tp_context: int, buffer: int, source, broadcast_type: int, targets, dim: int, t_device: int = -1) -> None:
    """ext_tp.cpp:129-180: source -> every target (one per device of the split).  A host source is the staging buffer
    pinned_temp[buffer] (or is copied into it first) and goes up to each device on that device's stream, as in the reference.
    A DEVICE source is copied to pinned_temp[buffer] too (the contract of the buffer), but the targets are written directly
    from the source over xGMI on their own streams -- no host bounce on the path the next kernel waits for."""
    ctx = _ctx(tp_context)
    if t_device != -1:
        raise NotImplementedError("tp_broadcast: per-thread form (TP_MULTITHREADED) is compiled out in the reference too")
    rows_cols = (source.numel() // source.shape[-1], source.shape[-1])
    pin = _pinned_view(ctx, buffer, *rows_cols, dtype=source.dtype)
    src2 = source.reshape(rows_cols)
    produced = None
    if source.device.type == "cuda":
        sdev = source.device.index
        with ctx.on(sdev):
            produced = torch.cuda.Event()
            produced.record(ctx.streams[sdev])
            if src2.data_ptr() != source.data_ptr():                # (reshape had to copy: keep the copy alive for the side streams)
                for d in ctx.all_devices:
                    src2.record_stream(ctx.streams[d])
            _e.copy_2d_async(pin, src2, ctx.streams[sdev].cuda_stream)          # D2H, source device's stream
    elif source.data_ptr() != pin.data_ptr():
        # a host tensor that is not the staging buffer yet: earlier uploads FROM the buffer may still be in flight on the
        # device streams -- they must have read it before it is overwritten
        if not ctx.host_only:
            for d in ctx.all_devices:
                ctx.streams[d].synchronize()
        pin.copy_(src2)
    for i, (dev, _, _) in enumerate(ctx.split(broadcast_type)):
        tgt = targets[i]
        if tgt.data_ptr() == source.data_ptr():
            continue
        if ctx.host_only:
            tgt.view(rows_cols).copy_(pin)
            continue
        with ctx.on(dev):
            if produced is not None:
                ctx.streams[dev].wait_event(produced)
                _e.copy_2d_async(tgt.view(rows_cols), src2, ctx.streams[dev].cuda_stream)      # peer copy, target's stream
            else:
                _e.copy_2d_async(tgt.view(rows_cols), pin, ctx.streams[dev].cuda_stream)       # H2D from the pinned buffer
    tp_cross_device_barrier(tp_context, broadcast_type, t_device)


def tp_gather(tp_context: int, buffer: int, inputs, broadcast_type: int, targets, broadcast_type_target: int, dim: int,
              t_device: int = -1) -> None:
    """ext_tp.cpp:182-287: column slices [rows, (b - a) * dim] of the split's devices form one [rows, last * dim] matrix.
    broadcast_type_target < 0: the matrix is wanted on the HOST (pinned_temp[buffer]; the reference's ctx.gather reads it
    there): one strided 2-D copy per slice on its device's stream (hipMemcpy2DAsync: asynchronous also when rows > 1 and the
    destination rows are not contiguous).  broadcast_type_target >= 0: it is wanted on the devices of the target split
    (ctx.allgather): every target pulls every slice directly over xGMI on its own stream -- the reference's round trip
    through the host buffer (one copy down per source, one up per target) is not made."""
    ctx = _ctx(tp_context)
    if t_device != -1:
        raise NotImplementedError("tp_gather: per-thread form (TP_MULTITHREADED) is compiled out in the reference too")
    split = ctx.split(broadcast_type)
    rows = inputs[0].shape[0]
    cols = split[-1][2] * dim
    pin = _pinned_view(ctx, buffer, rows, cols, dtype=inputs[0].dtype)
    if ctx.host_only:
        for i, (dev, a, _) in enumerate(split):
            pin[:, a * dim:a * dim + inputs[i].shape[1]].copy_(inputs[i])
        if broadcast_type_target >= 0:
            for i, (dev, _, _) in enumerate(ctx.split(broadcast_type_target)):
                targets[i].view(rows, cols).copy_(pin)
        return
    if broadcast_type_target < 0:
        for i, (dev, a, _) in enumerate(split):
            src = inputs[i]
            with ctx.on(dev):
                _e.copy_2d_async(pin[:, a * dim:a * dim + src.shape[1]], src, ctx.streams[dev].cuda_stream)
        if broadcast_type_target == -2:
            return
        tp_cross_device_barrier(tp_context, broadcast_type, t_device)
        return
    # device targets: slice i is complete on its device's stream at this point
    for dev, _, _ in split:
        ctx.record(dev)
    for t, (tdev, _, _) in enumerate(ctx.split(broadcast_type_target)):
        tgt = targets[t].view(rows, cols)
        with ctx.on(tdev):
            for i, (dev, a, _) in enumerate(split):
                src = inputs[i]
                if dev != tdev:
                    ctx.wait(tdev, dev)
                _e.copy_2d_async(tgt[:, a * dim:a * dim + src.shape[1]], src, ctx.streams[tdev].cuda_stream)
    # the sources may be reused by their devices only after every target has pulled them
    tp_cross_device_barrier(tp_context, broadcast_type_target, t_device)


def tp_all_reduce(tp_context: int, buffer: int, tensors, residuals) -> None:
    """ext_tp.cpp:392-474: residuals[i] += tensors[i], chained through the host buffer; every device ends with the sum."""
    ctx = _ctx(tp_context)
    n = len(tensors)
    rows_cols = (tensors[0].numel() // tensors[0].shape[-1], tensors[0].shape[-1])
    pin = _pinned_view(ctx, buffer, *rows_cols, dtype=tensors[0].dtype)
    for i in range(n):
        dev = tensors[i].device.index
        with ctx.on(dev):
            if i > 0:
                ctx.wait(dev, tensors[i - 1].device.index)
                residuals[i].view(rows_cols).copy_(pin, non_blocking=True)
            residuals[i].add_(tensors[i])
            pin.copy_(residuals[i].view(rows_cols), non_blocking=True)
            ctx.record(dev)
    last = tensors[n - 1].device.index
    for i in range(n - 1):
        dev = tensors[i].device.index
        with ctx.on(dev):
            ctx.wait(dev, last)
            residuals[i].view(rows_cols).copy_(pin, non_blocking=True)


def gemm_half_q_half_tp(a, b, c, force_cuda: bool, tp_context: int, t_device: int = -1) -> None:
    """ext_qmatrix.cpp:249-289: c[i] = a[i] . W_i on the device that holds slice i."""
    ctx = _ctx(tp_context)
    for ai, handle, ci in zip(a, b, c):
        with ctx.on(ci.device.index):
            _e.gemm_half_q_half(ai, handle, ci, force_cuda)


def rms_norm_tp(x, w, y, epsilon: float, tp_context: int) -> None:
    """ext_norm.cpp:59-93"""
    ctx = _ctx(tp_context)
    for xi, wi, yi in zip(x, w, y):
        with ctx.on(xi.device.index):
            _e.rms_norm(xi, wi, yi, epsilon)


def _narrow(ts, rows):
    return [t.narrow(0, 0, rows) for t in ts]


def _add_residual(ctx, outs, bc0):
    """ext_qmlp.cpp:434-451 / ext_qattn.cpp:462-480: out_i += the matching column slice of the broadcast input"""
    offset = 0
    for o, r in zip(outs, bc0):
        with ctx.on(o.device.index):
            w = o.shape[1]
            o.add_(r.narrow(1, offset, w))
            offset += w


def tp_mlp_forward_(tp_context: int, hidden_states, temp_bc0, temp_bc1, temp_bc2, temp_gate, temp_up, temp_down,
                    pre_layernorm, norm_epsilon: float, gate, up, down, act_gelu: bool) -> None:
    """ext_qmlp.cpp:326-473.  Result: pinned_temp[0] = x + down(act(gate(n)) * up(n)), n = rmsnorm(x)."""
    ctx = _ctx(tp_context)
    rows = hidden_states.shape[0]
    bc0, bc1, bc2 = _narrow(temp_bc0, rows), _narrow(temp_bc1, rows), _narrow(temp_bc2, rows)
    tg, tu, td = _narrow(temp_gate, rows), _narrow(temp_up, rows), _narrow(temp_down, rows)
    tp_broadcast(tp_context, 0, hidden_states, BROADCAST_ID, bc0, 1, -1)
    if len(pre_layernorm):
        rms_norm_tp(bc0, pre_layernorm, bc1, norm_epsilon, tp_context)
    else:
        bc1 = bc0
    if isinstance(gate, (list, tuple)) and len(gate):
        gemm_half_q_half_tp(bc1, gate, tg, False, tp_context, -1)
        gemm_half_q_half_tp(bc1, up, tu, False, tp_context, -1)
        for g, u in zip(tg, tu):
            with ctx.on(g.device.index):
                _e.act_mul_(g, u, act_gelu)
    else:
        raise NotImplementedError("tp_mlp_forward_: MLP without a gate projection")
    tp_gather(tp_context, 1, tg, BROADCAST_ID, bc2, BROADCAST_ID, 1, -1)
    gemm_half_q_half_tp(bc2, down, td, False, tp_context, -1)
    _add_residual(ctx, td, bc0)
    tp_gather(tp_context, 0, td, BROADCAST_RS, td, -1, 1, -1)


def _attn_common(tp_context, hidden_states, temp_bc0, temp_bc1, temp_bc2, temp_q, temp_k, temp_v, temp_o, k_cache, v_cache,
                 pre_layernorm, norm_epsilon, q_proj, k_proj, v_proj, o_proj, head_dim, rope_style, batch_size, q_len,
                 sin, cos, past_lens, block_index, scaling):
    ctx = _ctx(tp_context)
    rows = batch_size * q_len
    bc0, bc1, bc2 = _narrow(temp_bc0, rows), _narrow(temp_bc1, rows), _narrow(temp_bc2, rows)
    tq, tk, tv, to = _narrow(temp_q, rows), _narrow(temp_k, rows), _narrow(temp_v, rows), _narrow(temp_o, rows)
    tp_broadcast(tp_context, 0, hidden_states, BROADCAST_Q, bc0, head_dim, -1)
    if len(pre_layernorm):
        rms_norm_tp(bc0, pre_layernorm, bc1, norm_epsilon, tp_context)
    else:
        bc1 = bc0
    gemm_half_q_half_tp(bc1, q_proj, tq, False, tp_context, -1)
    gemm_half_q_half_tp(bc1, k_proj, tk, False, tp_context, -1)
    gemm_half_q_half_tp(bc1, v_proj, tv, False, tp_context, -1)
    for i, (q2, k2, v2, o2) in enumerate(zip(tq, tk, tv, to)):
        dev = ctx.split(BROADCAST_Q)[i][0]                 # sin / cos are indexed by device id (tensor_p.py:384-397)
        nh, nkv = q2.shape[1] // head_dim, k2.shape[1] // head_dim
        with ctx.on(dev):
            if rope_style != ROPE_STYLE_NONE:
                # rope_cuda_qk with past_len = 0 + per-sequence past_lens (ext_qattn.cpp:373-390): q and k in place
                for t, h in ((q2, nh), (k2, nkv)):
                    _e.rope_(t.view(batch_size, q_len, h, head_dim), sin[dev], cos[dev], 0, h, head_dim, past_lens[i],
                             rope_style == ROPE_STYLE_NEOX)
            q4 = q2.view(batch_size, q_len, nh, head_dim)
            k4 = k2.view(batch_size, q_len, nkv, head_dim)
            v4 = v2.view(batch_size, q_len, nkv, head_dim)
            # flash_attn_2_cuda.fwd_kvcache(q, k_cache, v_cache, k, v, cache_seqlens, ..., block_table, ..., out, scale,
            # causal = True) (ext_qattn.cpp:416-438 / 653-675): append k, v at cache_seqlens, attend, write `out`
            bt = None if block_index is None else block_index[i]
            o = _e.flash_attn_with_kvcache(q4, k_cache[i], v_cache[i], k4, v4, past_lens[i], bt, True, scaling)
            o2.view(batch_size, q_len, nh, head_dim).copy_(o)
    tp_gather(tp_context, 1, to, BROADCAST_Q, bc2, BROADCAST_Q, head_dim, -1)
    gemm_half_q_half_tp(bc2, o_proj, to, False, tp_context, -1)
    _add_residual(ctx, to, bc0)
    tp_gather(tp_context, 0, to, BROADCAST_Q, to, -1, head_dim, -1)


def tp_attn_forward_paged_(tp_context, hidden_states, temp_bc0, temp_bc1, temp_bc2, temp_q, temp_k, temp_v, temp_o, k_cache,
                           v_cache, pre_layernorm, norm_epsilon, q_proj, k_proj, v_proj, o_proj, head_dim, rope_style,
                           batch_size, q_len, sin, cos, past_lens, block_index, scaling) -> None:
    """ext_qattn.cpp:261-497.  Result in pinned_temp[0]."""
    _attn_common(tp_context, hidden_states, temp_bc0, temp_bc1, temp_bc2, temp_q, temp_k, temp_v, temp_o, k_cache, v_cache,
                 pre_layernorm, norm_epsilon, q_proj, k_proj, v_proj, o_proj, head_dim, rope_style, batch_size, q_len,
                 sin, cos, past_lens, block_index, scaling)


def tp_attn_forward_(tp_context, hidden_states, temp_bc0, temp_bc1, temp_bc2, temp_q, temp_k, temp_v, temp_o, k_cache, v_cache,
                     pre_layernorm, norm_epsilon, q_proj, k_proj, v_proj, o_proj, head_dim, rope_style, batch_size, q_len,
                     sin, cos, past_len_tp, scaling) -> None:
    """ext_qattn.cpp:499-700: the same over a contiguous cache [batch, max_seq_len, kv_heads_i, head_dim] (no block table)."""
    _attn_common(tp_context, hidden_states, temp_bc0, temp_bc1, temp_bc2, temp_q, temp_k, temp_v, temp_o, k_cache, v_cache,
                 pre_layernorm, norm_epsilon, q_proj, k_proj, v_proj, o_proj, head_dim, rope_style, batch_size, q_len,
                 sin, cos, past_len_tp, None, scaling)
