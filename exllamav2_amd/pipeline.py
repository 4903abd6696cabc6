"""Layer-split ("gpu_split", reference model.py:176-263) decode across ranks, one process per GPU.

The reference keeps one process driving all GPUs and copies the hidden state device -> device at every split point
(model.py:1014-1016, compat.py:53-139); only one GPU works at a time.  Here rank r owns layers [r L/N, (r+1) L/N) and
N independent sequences are in flight, one per stage: at every tick each rank runs ITS layers on the sequence currently
at its stage and hands the [hidden] fp16 vector to rank r+1 with one RCCL point-to-point send (xGMI is a point-to-point
fabric: a ring of sends uses one link per hop, no collective is needed).  The last rank samples (greedy) and sends the
token id to rank 0 in the same fixed-size message.  Per-GPU work per token is constant in N (weak scaling).

Works with any torch.distributed backend: nccl (= RCCL) on GPUs, gloo in the CPU tests (with the emulation library).
"""
from __future__ import annotations

import os
import time

import torch
import torch.distributed as dist

from .cache import ExLlamaV2Cache, ExLlamaV2Cache_Q4, PAGE_SIZE
from .model import ExLlamaV2
from .synth import synth_checkpoint


def split_layers(n_layers: int, world: int, rank: int):
    a = rank * n_layers // world
    b = (rank + 1) * n_layers // world
    return list(range(a, b))


class PipelineStage:
    """One rank's slice of the model + per-sequence decode state (device-side positions, like GreedyGraphDecoder)."""

    def __init__(self, cfg, rank: int, world: int, device, n_seqs: int, max_seq_len: int, recipe: str = "4.0bpw",
                 seed: int = 0, ext=None, use_graph: bool = True, cache_type: str = "fp16"):
        self.cfg, self.rank, self.world, self.n_seqs = cfg, rank, world, n_seqs
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
        self.msg_in = torch.zeros((h,), dtype=torch.float16, device=dev)
        self.msg_out = torch.zeros((h,), dtype=torch.float16, device=dev)
        self.x = torch.zeros((1, 1, h), dtype=torch.float16, device=dev)
        self.xn = torch.zeros((1, h), dtype=torch.float16, device=dev)
        self.ids = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.logits = torch.zeros((1, self.model.vocab_padded), dtype=torch.float16, device=dev)
        self.history = torch.zeros((n_seqs, max_seq_len + 2), dtype=torch.int32, device=dev)
        self.use_graph = use_graph and self.device.type == "cuda"
        self.stream = torch.cuda.Stream(device=dev) if self.device.type == "cuda" else None
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

    # one stage step for sequence s: msg_in -> (layers) -> msg_out
    def _step_eager(self, s: int):
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

    def exchange(self):
        """msg_out -> next rank, msg_in <- previous rank (ring; last -> first carries the sampled token).  Issued on the
        stage's own stream, so compute and hand-off are ordered without host synchronisation (RCCL: wait() is a stream
        wait; gloo: it blocks the host)."""
        nxt, prv = (self.rank + 1) % self.world, (self.rank - 1) % self.world
        with self._on_stream():
            ops = [dist.P2POp(dist.isend, self.msg_out, nxt), dist.P2POp(dist.irecv, self.msg_in, prv)]
            for r in dist.batch_isend_irecv(ops):
                r.wait()

    def free(self):
        for g in self.graphs.values():
            self.ext.graph_free(g)
        self.graphs = {}


def run_pipeline(stage: PipelineStage, first_tokens, n_ticks: int):
    """Drive `n_ticks` ticks.  At tick t rank r works on sequence (t - r) mod n_seqs (idle while the pipe fills).
    Sequence s starts from token first_tokens[s].  Returns the number of tokens sampled by the last rank."""
    n = stage.n_seqs
    sampled = 0
    for t in range(n_ticks):
        s = (t - stage.rank) % n
        active = t >= stage.rank
        if stage.first and t < n:
            # inject the prompt token of sequence t (instead of a token coming back from the last rank)
            with stage._on_stream():
                stage.msg_in.zero_()
                stage.msg_in[:2].view(torch.int32).copy_(torch.tensor([int(first_tokens[t])], dtype=torch.int32).to(stage.device))
        if active:
            stage.step(s)
            if stage.last:
                sampled += 1
        stage.exchange()
    return sampled


def run_layer_split_bench(cfg, args, rank: int, world: int, device, ext=None):
    """bench.py backend for --gpus N > 1: N sequences in flight over an N-stage layer split."""
    n_seqs = world
    on_gpu = torch.device(device).type == "cuda"
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    ramp = getattr(args, "ramp", 2048)   # untimed ticks before the warm-up (clock ramp of a freshly started GPU), same on every rank
    max_seq = max(2048, ((args.ctx + (args.steps + args.warmup + ramp) // n_seqs + 2 + 255) // 256) * 256)
    t_load = time.perf_counter()
    stage = PipelineStage(cfg, rank, world, device, n_seqs, max_seq, recipe=args.recipe, use_graph=not args.no_graph,
                          cache_type=getattr(args, "cache", "fp16"), ext=ext)
    stage.capture()
    stage.seqlens.fill_(args.ctx)
    sync()
    t_load = time.perf_counter() - t_load
    first = list(range(1, n_seqs + 1))
    # fill the pipe + warm-up (a "step" = one token sampled somewhere in the pipe = one tick once the pipe is full)
    run_pipeline(stage, first, world + ramp + args.warmup)
    sync()
    dist.barrier()
    t0 = time.perf_counter()
    n = stage.n_seqs
    base = world + ramp + args.warmup
    for t in range(base, base + args.steps):
        stage.step((t - rank) % n)
        stage.exchange()
    sync()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    # self-check for a driver-run scaling line: every rank reports what it holds and how many tokens its last stage sampled
    wb = torch.tensor([float(stage.model.weight_bytes())], dtype=torch.float64, device=device)
    gathered = [torch.zeros_like(wb) for _ in range(world)]
    dist.all_gather(gathered, wb)
    stage.free()
    return {"value": args.steps / dt, "ms_per_step": dt / args.steps * 1e3, "load_s": t_load,
            "parallelism": f"layer-split pipeline x{world} ({dist.get_backend()} ranks: {dist.get_world_size()}), "
                           f"{n_seqs} sequences in flight, {cfg.num_hidden_layers // world}-{-(-cfg.num_hidden_layers // world)} layers per rank",
            "weight_bytes_per_rank": [int(g.item()) for g in gathered]}
