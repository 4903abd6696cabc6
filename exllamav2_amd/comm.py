"""The two exchanges of the multi-GPU paths -- a ring hand-off (pipeline.py) and an all-gather of column pieces (tensor_p.py) --
behind one switch: on RCCL (backend "nccl") they are device-to-device over xGMI and stream-ordered; on gloo with DEVICE tensors
(a GPU test box with fewer GPUs than ranks: several ranks share cuda:0, which RCCL refuses) the payload is staged through host
memory, so the schedules, buffers and kernels of the real library run unchanged and only the wire differs.  gloo with host
tensors (the CPU emulation tests) passes straight through.

Reference: ext_tp.cpp:129-287 (tp_broadcast / tp_gather through pinned host memory), model.py:1014-1016 (split-point copy)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def host_staged(t: torch.Tensor, group=None) -> bool:
    return t.is_cuda and dist.get_backend(group) == "gloo"


def ring_exchange(out: torch.Tensor, inp: torch.Tensor, nxt: int, prv: int, group=None):
    """out -> rank nxt, inp <- rank prv.  Device tensors on RCCL: batched isend / irecv on the current stream (wait() is a stream
    wait).  Host-staged: the current stream is drained first (the message must have been produced), the receive lands in `inp`
    with a blocking copy."""
    if not host_staged(out, group):
        ops = [dist.P2POp(dist.isend, out, nxt, group), dist.P2POp(dist.irecv, inp, prv, group)]
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        return
    torch.cuda.current_stream(out.device).synchronize()
    h_out, h_in = out.cpu(), torch.empty(inp.shape, dtype=inp.dtype)
    ops = [dist.P2POp(dist.isend, h_out, nxt, group), dist.P2POp(dist.irecv, h_in, prv, group)]
    for r in dist.batch_isend_irecv(ops):
        r.wait()
    inp.copy_(h_in)


def all_gather(pieces: list, local: torch.Tensor, group=None):
    """pieces[r] <- rank r's `local` (equal shapes)."""
    if not host_staged(local, group):
        dist.all_gather(pieces, local, group=group)
        return
    torch.cuda.current_stream(local.device).synchronize()
    h = [torch.empty(local.shape, dtype=local.dtype) for _ in pieces]
    dist.all_gather(h, local.cpu(), group=group)
    for p, s in zip(pieces, h):
        p.copy_(s)


def all_reduce_sum(t: torch.Tensor, group=None):
    """t <- sum over the ranks of t, in place (fp32 payload: the partial sums of a row-parallel linear, tensor_p.py)."""
    if not host_staged(t, group):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return
    torch.cuda.current_stream(t.device).synchronize()
    h = t.cpu()
    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
    t.copy_(h)
