"""Layer-split ("gpu_split", reference model.py:176-263) decode across ranks, one process per GPU.

The reference keeps one process driving all GPUs and copies the hidden state device -> device at every split point
(model.py:1014-1016, compat.py:53-139); only one GPU works at a time.  Here rank r owns layers [r L/N, (r+1) L/N) and
`depth` x N independent sequences are in flight: at every tick each rank runs ITS layers on the sequence currently
at its stage and hands the [hidden] fp16 vector to rank r+1 with one RCCL point-to-point send (xGMI is a point-to-point
fabric: a ring of sends uses one link per hop, no collective is needed).  The last rank samples (greedy) and sends the
token id to rank 0 in the same fixed-size message.  Per-GPU work per token is constant in N (weak scaling).

Hand-off (round 4; BASELINE north_star: "async copies overlapped on HIP streams").  depth = 1 is the lock-step form: one
message buffer, the send / receive pair of tick t on the stage's compute stream, so a tick costs compute + hand-off.
depth = 2 (default for N > 1) gives every message a whole tick to travel: 2 N sequences in flight, sequence s sits at
rank r at ticks t = s + 2 r (mod 2 N), message buffers are double (index = parity of the tick = parity of the sequence,
so the per-sequence graphs keep fixed addresses), and the send / receive pair of tick t runs on a SECOND stream behind an
event of compute(t) while the compute stream goes straight on to tick t + 1 (the other buffer pair); compute(t + 2) waits
for the event of that exchange.  A tick then costs max(compute, hand-off).

Works with any torch.distributed backend: nccl (= RCCL) on GPUs, gloo in the CPU tests (with the emulation library).
"""
from __future__ import annotations

import os
import time

import torch
import torch.distributed as dist

from .cache import ExLlamaV2Cache, ExLlamaV2Cache_Q4, PAGE_SIZE
from .comm import ring_exchange
from .model import ExLlamaV2
from .synth import synth_checkpoint


def split_layers(n_layers: int, world: int, rank: int):
    a = rank * n_layers // world
    b = (rank + 1) * n_layers // world
    return list(range(a, b))


class PipelineStage:
    """One rank's slice of the model + per-sequence decode state (device-side positions, like GreedyGraphDecoder)."""

    def __init__(self, cfg, rank: int, world: int, device, n_seqs: int, max_seq_len: int, recipe: str = "4.0bpw",
                 seed: int = 0, ext=None, use_graph: bool = True, cache_type: str = "fp16", depth: int = 1):
        assert depth in (1, 2) and (depth == 1 or n_seqs % 2 == 0), "depth 2 needs an even number of sequences"
        self.cfg, self.rank, self.world, self.n_seqs, self.depth = cfg, rank, world, n_seqs, depth
        self.first, self.last = rank == 0, rank == world - 1
        self.device = torch.device(device)
        layers = split_layers(cfg.num_hidden_layers, world, rank)
        ck = synth_checkpoint(cfg, device, recipe=recipe, seed=seed, layers=layers, with_embed=self.first,
                              with_head=self.last)
        self.model = ExLlamaV2(cfg, device=device, ext=ext).load(ck, layers=layers)
        self.ext = self.model.ext
        cache_cls = ExLlamaV2Cache_Q4 if cache_type == "q4" else ExLlamaV2Cache
        self.cache = cache_cls(self.model, batch_size=n_seqs, max_seq_len=max_seq_len)
        pages = max_seq_len // PAGE_SIZE
        dev = self.device
        self.block_table = torch.arange(n_seqs * pages, dtype=torch.int32, device=dev).view(n_seqs, pages).contiguous()
        self.seqlens = torch.zeros((n_seqs,), dtype=torch.int32, device=dev)
        h = cfg.hidden_size
        # message = [hidden] fp16; the last -> first hop carries the token id in its first 4 bytes
        # (depth 2: one pair per tick parity; self.msg_in / self.msg_out always name the pair of the sequence being stepped)
        self.msg_in_buf = [torch.zeros((h,), dtype=torch.float16, device=dev) for _ in range(depth)]
        self.msg_out_buf = [torch.zeros((h,), dtype=torch.float16, device=dev) for _ in range(depth)]
        self.msg_in, self.msg_out = self.msg_in_buf[0], self.msg_out_buf[0]
        self.x = torch.zeros((1, 1, h), dtype=torch.float16, device=dev)
        self.xn = torch.zeros((1, h), dtype=torch.float16, device=dev)
        self.ids = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.logits = torch.zeros((1, self.model.vocab_padded), dtype=torch.float16, device=dev)
        self.history = torch.zeros((n_seqs, max_seq_len + 2), dtype=torch.int32, device=dev)
        self.use_graph = use_graph and self.device.type == "cuda"
        self.stream = torch.cuda.Stream(device=dev) if self.device.type == "cuda" else None
        # depth 2: the hand-off's own stream and, per buffer pair, "compute done" / "exchange done" events
        self.comm_stream = torch.cuda.Stream(device=dev) if (self.device.type == "cuda" and depth == 2) else None
        self.ev_compute = [torch.cuda.Event() for _ in range(depth)] if self.comm_stream is not None else None
        self.ev_comm = [torch.cuda.Event() for _ in range(depth)] if self.comm_stream is not None else None
        self.graphs = {}
        self.pos = [0] * n_seqs                                   # host mirror of seqlens: the kernels do not bound-check
        self.limit = min(max_seq_len, cfg.max_seq_len)
        self.chain = None
        if os.environ.get("EXL2_CHAIN", "1") != "0":
            self._setup_chain()

    def _setup_chain(self):
        """The chained decode route of model.GreedyGraphDecoder (csrc/qgemv_flat.hip: producers leave the residual stream in
        the consumer's packed order + partial sums of squares) inside a stage.  Same conditions: dense chain-capable layers,
        FP16 cache.  What differs at a stage boundary: a stage that is not the first receives x in natural order and
        PUBLISHES it for its first layer with the embedding kernel reading the one-row table `msg_in` (id 0); a stage that is
        not the last hands over the natural-order copy every chained epilogue also writes."""
        m, ext, cfg = self.model, self.ext, self.cfg
        if getattr(self.cache, "wbits", 0) or not m.layers:
            return
        plan = []
        for attn, mlp in m.layers:
            if not hasattr(mlp, "gate_proj") or mlp.q_handle is None or attn.q_handle is None:
                return
            cap_a, in_a, o_inv, nw_a = ext.q_attn_chain_info(attn.q_handle)
            cap_m, in_m, nw_m = ext.q_mlp_chain_info(mlp.q_handle)
            if not (cap_a and cap_m):
                return
            plan.append((in_a, o_inv, in_m, nw_a, nw_m))
        dev, h = self.device, cfg.hidden_size
        ch = {"plan": plan,
              "xp_a": torch.zeros((1, h), dtype=torch.float16, device=dev), "xp_b": torch.zeros((1, h), dtype=torch.float16, device=dev),
              "ss_a": torch.zeros((1, 512), dtype=torch.float32, device=dev), "ss_b": torch.zeros((1, 512), dtype=torch.float32, device=dev),
              "id0": torch.zeros((1,), dtype=torch.int32, device=dev)}
        if self.last:
            if m.lm_head is None:
                return
            head_perm, head_inv = ext.q_matrix_perm_info(m.lm_head.q_handle)
            ch["head_inv"] = head_inv
            ch["norm_head"] = torch.empty_like(m.norm.weight)
            ext.gather_f16(m.norm.weight, head_perm, ch["norm_head"])
        self.chain = ch

    def _step_chain(self, s: int):
        m, ext, cfg, ch = self.model, self.ext, self.cfg, self.chain
        plan = ch["plan"]
        x2 = self.x.view(1, cfg.hidden_size)
        q = m.temp_q[:1].view(1, 1, cfg.num_attention_heads, cfg.head_dim)
        k = m.temp_k[:1].view(1, 1, cfg.num_key_value_heads, cfg.head_dim)
        v = m.temp_v[:1].view(1, 1, cfg.num_key_value_heads, cfg.head_dim)
        xp_a, xp_b, ss_a, ss_b = ch["xp_a"], ch["xp_b"], ch["ss_a"], ch["ss_b"]
        if self.first:
            self.ids.copy_(self.msg_in[:2].view(torch.int32))
            ext.embed_rows_chain(m.embed_tokens, self.ids, x2, plan[0][0], plan[0][3], xp_a, ss_a)
        else:
            ext.embed_rows_chain(self.msg_in.view(1, cfg.hidden_size), ch["id0"], x2, plan[0][0], plan[0][3], xp_a, ss_a)
        sl, bt = self.seqlens[s:s + 1], self.block_table[s:s + 1]
        npart = 1
        for i, (attn, mlp) in enumerate(m.layers):
            in_a, o_inv, in_m, nw_a, nw_m = plan[i]
            ext.q_attn_forward_1_chain(attn.q_handle, xp_a, ss_a, npart, 1, q, k, v)
            ao = attn.attend_chain(q, k, v, self.cache, sl, bt, o_inv)
            npart = ext.q_attn_forward_2_chain(attn.q_handle, x2, ao, 1, in_m, nw_m, xp_b, ss_b)
            # behind the stage's last layer of a non-last stage nothing reads the packed copy: any layer's order will do
            if i + 1 < len(plan): nxt, nxt_w = plan[i + 1][0], plan[i + 1][3]
            elif self.last: nxt, nxt_w = ch["head_inv"], ch["norm_head"]
            else: nxt, nxt_w = plan[0][0], plan[0][3]
            npart = ext.q_mlp_forward_chain(mlp.q_handle, x2, xp_b, ss_b, npart, 1, nxt, nxt_w, xp_a, ss_a)
        ext.add_i32_(sl, 1)
        if self.last:
            ext.gemm_half_q_half_chain(xp_a, ss_a, npart, cfg.norm_eps, m.lm_head.q_handle, self.logits, 1)
            ext.argmax_rows(self.logits, self.ids, cfg.vocab_size, self.history[s:s + 1], sl)
            self.msg_out.zero_()
            self.msg_out[:2].view(torch.int32).copy_(self.ids)
        else:
            self.msg_out.copy_(self.x.view(-1))

    def _select_buffers(self, s: int):
        p = s % self.depth
        self.msg_in, self.msg_out = self.msg_in_buf[p], self.msg_out_buf[p]
        return p

    # one stage step for sequence s: msg_in -> (layers) -> msg_out
    def _step_eager(self, s: int):
        self._select_buffers(s)
        if self.chain is not None:
            try:
                return self._step_chain(s)
            except RuntimeError as e:
                if "not covered" not in str(e):
                    raise
                self.chain = None               # a shape outside the chained kernels: the module-by-module route below
        m, ext, cfg = self.model, self.ext, self.cfg
        if self.first:
            self.ids.copy_(self.msg_in[:2].view(torch.int32))
            ext.embed_rows(m.embed_tokens, self.ids, self.x.view(1, cfg.hidden_size))
        else:
            self.x.view(-1).copy_(self.msg_in)
        sl, bt = self.seqlens[s:s + 1], self.block_table[s:s + 1]
        for attn, mlp in m.layers:
            # cache rows of sequence s through its own block-table row (pages of batch row s)
            attn.forward(self.x, self.cache, 0, sl, bt)
            mlp.forward(self.x)
        ext.add_i32_(sl, 1)
        if self.last:
            ext.rms_norm(self.x.view(1, -1), m.norm.weight, self.xn, cfg.norm_eps)
            ext.gemm_half_q_half(self.xn, m.lm_head.q_handle, self.logits)
            ext.argmax_rows(self.logits, self.ids, cfg.vocab_size, self.history[s:s + 1], sl)
            self.msg_out.zero_()
            self.msg_out[:2].view(torch.int32).copy_(self.ids)
        else:
            self.msg_out.copy_(self.x.view(-1))

    def capture(self):
        if not self.use_graph:
            return self
        torch.cuda.synchronize()
        with torch.cuda.stream(self.stream):
            saved = self.seqlens.clone()
            for s in range(self.n_seqs):
                self.seqlens.fill_(self.limit - 1)          # throw-away steps land in the last cache slot
                self._step_eager(s)
                self.seqlens.fill_(self.limit - 1)
                self.stream.synchronize()
                self.ext.graph_begin_capture(self.stream.cuda_stream)
                try:
                    self._step_eager(s)
                finally:
                    self.graphs[s] = self.ext.graph_end_capture(self.stream.cuda_stream)
            self.seqlens.copy_(saved)
            self.stream.synchronize()
        return self

    def _on_stream(self):
        import contextlib
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def step(self, s: int):
        if self.pos[s] + 1 > self.limit:
            raise RuntimeError(f"pipeline stage: sequence {s} is at the cache / max_seq_len limit ({self.limit})")
        self.pos[s] += 1
        with self._on_stream():
            if self.use_graph:
                self.ext.graph_launch(self.graphs[s], self.stream.cuda_stream)
            else:
                self._step_eager(s)

    def begin_tick(self, p: int):
        """compute stream: the exchange that filled buffer pair p (two ticks ago) has completed"""
        if self.comm_stream is not None:
            self.stream.wait_event(self.ev_comm[p])

    def exchange(self, p: int = 0):
        """msg_out[p] -> next rank, msg_in[p] <- previous rank (ring; last -> first carries the sampled token).
        depth 1: issued on the stage's compute stream, so compute and hand-off are ordered without host synchronisation
        (RCCL: wait() is a stream wait; gloo: it blocks the host).  depth 2 on a GPU: issued on the hand-off stream behind
        an event of this tick's compute -- the compute stream does not wait; the next use of pair p (two ticks on) does."""
        nxt, prv = (self.rank + 1) % self.world, (self.rank - 1) % self.world
        if self.comm_stream is None:
            with self._on_stream():
                ring_exchange(self.msg_out_buf[p], self.msg_in_buf[p], nxt, prv)
            return
        self.ev_compute[p].record(self.stream)
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(self.ev_compute[p])      # msg_out[p] produced, msg_in[p] consumed
            ring_exchange(self.msg_out_buf[p], self.msg_in_buf[p], nxt, prv)
            self.ev_comm[p].record(self.comm_stream)

    def free(self):
        for g in self.graphs.values():
            self.ext.graph_free(g)
        self.graphs = {}


def tick(stage: PipelineStage, t: int, first_tokens=None) -> bool:
    """One tick of the schedule on this rank: sequence s = (t - depth * rank) mod n_seqs is stepped when the pipe has reached
    this stage, then the buffer pair of this tick is exchanged.  Returns True when the last stage sampled a token."""
    n, d = stage.n_seqs, stage.depth
    s = (t - d * stage.rank) % n
    p = t % d
    active = t >= d * stage.rank
    stage.begin_tick(p)
    if stage.first and first_tokens is not None and t < n:
        # inject the prompt token of sequence t (instead of a token coming back from the last rank)
        with stage._on_stream():
            stage.msg_in_buf[p].zero_()
            stage.msg_in_buf[p][:2].view(torch.int32).copy_(torch.tensor([int(first_tokens[t])], dtype=torch.int32).to(stage.device))
    if active:
        stage.step(s)
    stage.exchange(p)
    return active and stage.last


def run_pipeline(stage: PipelineStage, first_tokens, n_ticks: int, t0: int = 0):
    """Drive ticks t0 .. t0 + n_ticks.  At tick t rank r works on sequence (t - depth * r) mod n_seqs (idle while the pipe
    fills).  Sequence s starts from token first_tokens[s].  Returns the number of tokens sampled by the last rank."""
    sampled = 0
    for t in range(t0, t0 + n_ticks):
        sampled += int(tick(stage, t, first_tokens))
    if stage.comm_stream is not None:
        stage.stream.wait_stream(stage.comm_stream)            # whoever synchronises the compute stream sees the hand-offs done
    return sampled


def run_layer_split_bench(cfg, args, rank: int, world: int, device, ext=None):
    """bench.py backend for --gpus N > 1: depth x N sequences in flight over an N-stage layer split (depth 2 = overlapped
    hand-off, EXL2_PIPELINE_DEPTH=1 = the lock-step form).  A "step" = one tick = one token sampled by the last stage once the
    pipe is full; exactly args.steps ticks are timed between barriers."""
    depth = int(os.environ.get("EXL2_PIPELINE_DEPTH", "2"))
    n_seqs = depth * world
    on_gpu = torch.device(device).type == "cuda"
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    ramp = getattr(args, "ramp", 2048)   # untimed ticks before the warm-up (clock ramp of a freshly started GPU), same on every rank
    max_seq = max(2048, ((args.ctx + (args.steps + args.warmup + ramp) // n_seqs + 2 + 255) // 256) * 256)
    t_load = time.perf_counter()
    stage = PipelineStage(cfg, rank, world, device, n_seqs, max_seq, recipe=args.recipe, use_graph=not args.no_graph,
                          cache_type=getattr(args, "cache", "fp16"), ext=ext, depth=depth)
    stage.capture()
    stage.seqlens.fill_(args.ctx)
    sync()
    t_load = time.perf_counter() - t_load
    first = list(range(1, n_seqs + 1))
    # fill the pipe + warm-up
    base = depth * world + ramp + args.warmup
    run_pipeline(stage, first, base)
    sync()
    dist.barrier()
    t0 = time.perf_counter()
    run_pipeline(stage, None, args.steps, t0=base)
    sync()
    dist.barrier()
    tdev = "cpu" if dist.get_backend() == "gloo" else device      # (bookkeeping scalars: host tensors on gloo)
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=tdev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    # self-check for a driver-run scaling line: every rank reports what it holds
    wb = torch.tensor([float(stage.model.weight_bytes())], dtype=torch.float64, device=tdev)
    gathered = [torch.zeros_like(wb) for _ in range(world)]
    dist.all_gather(gathered, wb)
    per_rank = [int(g.item()) for g in gathered]
    stage.free()
    tick_s = dt / args.steps
    return {"value": args.steps / dt, "ms_per_step": tick_s * 1e3, "load_s": t_load,
            "sequences_in_flight": n_seqs,
            "parallelism": f"layer-split pipeline x{world} ({dist.get_backend()} ranks: {dist.get_world_size()}), "
                           f"{n_seqs} sequences in flight ({'double-buffered hand-off on a second stream' if depth == 2 else 'lock-step hand-off'}), "
                           f"{cfg.num_hidden_layers // world}-{-(-cfg.num_hidden_layers // world)} layers per rank",
            "weight_bytes_per_rank": per_rank,
            # every rank streams its own weights once per tick: fraction of ITS 8 TB/s
            "per_gpu_weight_roofline_frac": [round(b / tick_s / 8.0e12, 4) for b in per_rank]}
