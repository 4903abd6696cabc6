"""ExLlamaV2 model runtime for the Llama family (reference model.py:78-173 module list, :764-933 forward chunking,
:936-1054 forward_chunk) on one device, plus the whole-step greedy decode graph used by the benchmark."""
from __future__ import annotations

import math

import os

import torch

from .attn import ExLlamaV2Attention
from .cache import PAGE_SIZE
from .config import ExLlamaV2Config
from .ext import ext_c, none_tensor
from .linear import ExLlamaV2Linear
from .mlp import ExLlamaV2MLP
from .moe_mlp import ExLlamaV2MoEMLP
from .rmsnorm import ExLlamaV2RMSNorm


def rope_tables(cfg: ExLlamaV2Config, device):
    """device.py:118-169: fp16 sin/cos [max_seq_len, head_dim]."""
    inv_freq = 1.0 / (cfg.rotary_embedding_base ** (torch.arange(0, cfg.head_dim, 2, device=device).float() / cfg.head_dim))
    t = torch.arange(cfg.max_seq_len, device=device, dtype=torch.float32)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1) if cfg.rope_style == 2 else torch.repeat_interleave(freqs, 2, dim=-1)
    return emb.sin().half().contiguous(), emb.cos().half().contiguous()


class ExLlamaV2:
    _live_on_device: dict = {}          # loaded models per device (the library's staging scratch is per device and stream)
    def __init__(self, config: ExLlamaV2Config, device="cuda:0", ext=None):
        self.config = config
        self.device = torch.device(device)
        self.ext = ext or ext_c
        cfg = config
        # per-device scratch arena (device.py:102-115): sized for one forward chunk
        # rows of one forward chunk: max_input_len x max_batch_size, as the reference sizes its scratch (model.py:195,
        # attn.py:315, mlp.py:216) -- prefill GEMMs want the tallest chunk the arena allows (288 GB: 8 x 2048 rows of a 7B are 1 GB)
        self.max_rows = cfg.max_input_len * max(1, cfg.max_batch_size)
        r, dev = self.max_rows, self.device
        self.temp_state = torch.empty((r, cfg.hidden_size), dtype=torch.float16, device=dev)
        self.temp_a = torch.empty((r, cfg.intermediate_size), dtype=torch.float16, device=dev)
        self.temp_b = torch.empty((r, cfg.intermediate_size), dtype=torch.float16, device=dev)
        self.temp_q = torch.empty((r, cfg.num_attention_heads * cfg.head_dim), dtype=torch.float16, device=dev)
        self.temp_k = torch.empty((r, cfg.num_key_value_heads * cfg.head_dim), dtype=torch.float16, device=dev)
        self.temp_v = torch.empty((r, cfg.num_key_value_heads * cfg.head_dim), dtype=torch.float16, device=dev)
        self.temp_attn = torch.empty((r, cfg.num_attention_heads * cfg.head_dim), dtype=torch.float16, device=dev)
        sb = self.ext.paged_attn_scratch_bytes(min(r, 64) * cfg.num_attention_heads, cfg.head_dim, 64)
        self.attn_scratch = torch.empty((sb // 4 + 16,), dtype=torch.float32, device=dev)
        self.attn_counters = torch.zeros((16384,), dtype=torch.int32, device=dev)     # split hand-off tickets (attn.hip)
        self._dq = None                                   # reconstruct target of ExLlamaV2Linear.forward(force_recons=True)
        self.sin, self.cos = rope_tables(cfg, dev)
        self.modules = []
        self.layers = []
        self.embed_tokens = None
        self.norm = None
        self.lm_head = None
        self.loaded = False

    def dq_scratch(self, k: int, n: int, device) -> torch.Tensor:
        """fp16 [k, n] view of the per-device dequantisation scratch (reference: temp_dq, device.py:102-115); grows on demand"""
        if self._dq is None or self._dq.numel() < k * n:
            self._dq = torch.empty((k * n,), dtype=torch.float16, device=device)
        return self._dq[:k * n].view(k, n)

    def load(self, ck: dict, layers=None):
        """model.py:266-351: build the module handles from a checkpoint dict.  `layers` = the global layer indices this
        device owns (layer split, model.py:176-263); embedding / final norm + head are built when their tensors are in
        `ck`.  KV-cache slots are indexed by LOCAL layer position."""
        cfg = self.config
        self.embed_tokens = ck.get("model.embed_tokens")
        self.layer_ids = list(range(cfg.num_hidden_layers) if layers is None else layers)
        for local_idx, i in enumerate(self.layer_ids):
            key = f"model.layers.{i}"
            attn = ExLlamaV2Attention(self, key, local_idx).load(ck)
            if getattr(cfg, "num_experts", 0):
                mlp = ExLlamaV2MoEMLP(self, key, local_idx, cfg.num_experts, cfg.num_experts_per_token).load(ck)
            else:
                mlp = ExLlamaV2MLP(self, key, local_idx).load(ck)
            self.layers.append((attn, mlp))
            self.modules += [attn, mlp]
        vpad = (cfg.vocab_size + 31) // 32 * 32
        self.vocab_padded = vpad
        if "lm_head" in ck:
            self.norm = ExLlamaV2RMSNorm(self.ext, "model.norm", ck["model.norm"], cfg.norm_eps)
            self.lm_head = ExLlamaV2Linear(self.ext, "lm_head", cfg.hidden_size, vpad, self).load(ck["lm_head"])
        if not self.loaded:
            ExLlamaV2._live_on_device[str(self.device)] = ExLlamaV2._live_on_device.get(str(self.device), 0) + 1
        self.loaded = True
        return self

    def unload(self):
        for attn, mlp in self.layers:
            attn.unload(); mlp.unload()
        if self.lm_head: self.lm_head.unload()
        was_loaded = self.loaded
        self.layers, self.modules, self.loaded = [], [], False
        # the staging buffers of this device's streams (csrc/qgemm_prefill.hip: activations up to 360 MB per stream, decoded weight
        # fragments K x N x 2 bytes) are shared by every model of the device and their addresses sit in captured graphs (a decoder
        # of 5-16 sequences): they go when the LAST loaded model of the device goes -- a draft model unloading must not pull
        # them from under the main model's graphs
        if was_loaded:
            k = str(self.device)
            ExLlamaV2._live_on_device[k] = max(0, ExLlamaV2._live_on_device.get(k, 1) - 1)
            if ExLlamaV2._live_on_device[k] == 0:
                self.ext.release_scratch(self.device)

    def weight_bytes(self) -> int:
        """Algorithmic bytes one token streams through the linears.  A sparse-MoE layer streams
        its router plus `num_experts_per_token` of its experts per token (moe_mlp.py:238-253)."""
        n = self.lm_head.weight_bytes() if self.lm_head else 0
        for attn, mlp in self.layers:
            for lin in (attn.q_proj, attn.k_proj, attn.v_proj, attn.o_proj):
                n += lin.weight_bytes()
            if hasattr(mlp, "gate_proj"):
                n += sum(lin.weight_bytes() for lin in (mlp.gate_proj, mlp.up_proj, mlp.down_proj))
            else:
                per_expert = sum(l.weight_bytes() for l in mlp.w1 + mlp.w2 + mlp.w3) // mlp.num_experts
                n += per_expert * mlp.num_experts_per_token + mlp.gate.numel() * 2
        return n

    # ---- forward ------------------------------------------------------------------------------------------------------

    def forward_chunk(self, input_ids: torch.Tensor, cache, past_len: int = 0, cache_seqlens=None, block_table=None,
                      last_id_only: bool = True, preprocess_only: bool = False):
        """model.py:936-1054.  input_ids [b, q_len] (any device); returns logits [b, q_len | 1, vocab] or None."""
        cfg = self.config
        b, q_len = input_ids.shape
        assert b * q_len <= self.max_rows, "chunk larger than the scratch arena"
        x = self.embed_tokens[input_ids.to(self.device).reshape(-1)].view(b, q_len, cfg.hidden_size).contiguous()
        for attn, mlp in self.layers:
            attn.forward(x, cache, past_len, cache_seqlens, block_table)
            mlp.forward(x)
        if preprocess_only:
            return None
        if last_id_only:
            x = x[:, -1:, :].contiguous()
        x = self.norm.forward(x)
        logits = self.lm_head.forward(x)
        return logits[..., :cfg.vocab_size]

    def forward(self, input_ids: torch.Tensor, cache, last_id_only: bool = True, preprocess_only: bool = False,
                input_mask=None):
        """model.py:764-933: contiguous-cache forward with chunked prefill; advances cache.current_seq_len."""
        b, q_len = input_ids.shape
        chunk = max(1, self.max_rows // b)                    # model.py:833 effective_max_input_len
        past = cache.current_seq_len
        assert past + q_len <= cache.max_seq_len, "sequence exceeds cache"
        out = None
        pos = 0
        while pos < q_len:
            n = min(chunk, q_len - pos)
            last = pos + n == q_len
            r = self.forward_chunk(input_ids[:, pos:pos + n], cache, past + pos,
                                   last_id_only=last_id_only, preprocess_only=preprocess_only or (last_id_only and not last))
            if r is not None:
                out = r if (out is None or last_id_only) else torch.cat([out, r], dim=1)
            pos += n
        cache.current_seq_len = past + q_len
        return out


class _PartialSums:
    """Host-side layout of one buffer of partial sums of squares of the chained step.  A producer launch writes its rows packed,
    `[rows, P]` with P = the number of partials per row it published (known when the launch returns); a consumer launch reads
    `[rows, P]` from one base pointer.  Launches of a step may group the rows differently (csrc/qgemv_lean.hip ROWS form: the rows
    that fit in LDS depend on K), so groups are packed one behind the other and a consumer's rows must come out contiguous with one
    P -- which they do whenever the producer groups they span published the same P."""

    def __init__(self, buf: torch.Tensor, rows: int):
        self.buf, self.rows = buf.view(-1), rows
        self.pos, self.count = [0] * rows, [1] * rows
        self.next = 0

    def begin(self):
        """a producer pass over all rows starts"""
        self.next = 0

    def out(self, r0: int, r1: int) -> torch.Tensor:
        self._pending = (r0, r1, self.next)
        return self.buf[self.next:]

    def done(self, count: int):
        r0, r1, start = self._pending
        if start + (r1 - r0) * count > self.buf.numel():
            raise RuntimeError("chained decode: partial sums exceed their buffer: shape not covered by the chained decode kernel")
        for r in range(r0, r1):
            self.pos[r], self.count[r] = start + (r - r0) * count, count
        self.next = start + (r1 - r0) * count

    def inp(self, r0: int, r1: int, what: str):
        c = self.count[r0]
        if any(self.count[r] != c or self.pos[r] != self.pos[r0] + (r - r0) * c for r in range(r0, r1)):
            raise RuntimeError(f"chained decode: {what} rows {r0}..{r1 - 1} were produced with different numbers of partial sums: "
                               f"shape not covered by the chained decode kernel")
        return self.buf[self.pos[r0]:], c


class GreedyGraphDecoder:
    """One greedy decode step (embedding -> all layers -> norm -> head -> argmax -> advance positions) captured as a HIP
    graph.  Procedure = test_inference.py:604-609 (forward(ids[:, -1:], cache); argmax; append) with the token feedback
    and the position counters kept on the device, so a token costs one graph launch and no host synchronisation."""

    def __init__(self, model: ExLlamaV2, cache, batch_size: int = 1, max_new_tokens: int = 4096):
        self.model, self.cache, self.b = model, cache, batch_size
        cfg, dev = model.config, model.device
        pages_per_seq = cache.max_seq_len // PAGE_SIZE
        assert cache.max_seq_len % PAGE_SIZE == 0
        # contiguous cache rows seen through an identity block table (attn_params.py:241-257 "is_sequential")
        self.block_table = (torch.arange(batch_size * pages_per_seq, dtype=torch.int32, device=dev)
                            .view(batch_size, pages_per_seq).contiguous())
        self.cache_seqlens = torch.zeros((batch_size,), dtype=torch.int32, device=dev)
        self.ids = torch.zeros((batch_size,), dtype=torch.int32, device=dev)
        self.history = torch.zeros((batch_size, cache.max_seq_len + 2), dtype=torch.int32, device=dev)
        self.x = torch.zeros((batch_size, 1, cfg.hidden_size), dtype=torch.float16, device=dev)
        self.xn = torch.zeros_like(self.x)
        self.logits = torch.zeros((batch_size, model.vocab_padded), dtype=torch.float16, device=dev)
        self.graph = None
        # capture is not permitted on the legacy default stream: the decoder owns a stream (device.py:67-72 does too)
        self.stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self.chain = None
        self.pos = 0                  # host mirror of cache_seqlens (all rows advance together): bounds are checked here,
        self.limit = min(cache.max_seq_len, cfg.max_seq_len)      # the kernels index pages / sin-cos rows unchecked
        if os.environ.get("EXL2_CHAIN", "1") != "0":
            self._setup_chain()
        if os.environ.get("EXL2_CHAIN_OVERLAP", "0") != "0" and (self.chain is None or "flags" not in self.chain):
            raise RuntimeError("EXL2_CHAIN_OVERLAP=1 asked for the experimental overlapped chain, but this model / cache / batch is "
                               "not chain-capable (dense layers with shared q|k|v and gate|up permutations, FP16 cache, <= 16 rows)")

    def _setup_chain(self):
        """Chained decode (csrc/qgemv_flat.hip): every producer of the residual stream leaves it in its consumer's packed
        (act-order) order + partial sums of squares.  Needs: dense layers whose q/k/v and gate/up share a permutation (every
        quantizer-written checkpoint), an FP16 or Q4 paged-view cache, a head on this device, <= 16 rows."""
        m, ext, cfg = self.model, self.model.ext, self.model.config
        if self.b > 16 or m.lm_head is None or m.embed_tokens is None:
            return
        if getattr(self.cache, "wbits", 0) and not all(attn.q4_chain_capable(self.cache) for attn, _ in m.layers):
            return
        plan = []
        for attn, mlp in m.layers:
            if mlp.q_handle is None or attn.q_handle is None:
                return
            cap_a, in_a, o_inv, nw_a = ext.q_attn_chain_info(attn.q_handle)
            if hasattr(mlp, "gate_proj"):
                cap_m, in_m, nw_m = ext.q_mlp_chain_info(mlp.q_handle)
            elif hasattr(mlp, "num_experts") and os.environ.get("EXL2_CHAIN_MOE", "1") != "0" and "EXL2_CHAIN_OVERLAP" not in os.environ:
                # a sparse-MoE block (round 6): the attention half of the layer is chained (q|k|v from the published hand-off, o_proj
                # accumulating into x without publishing), the block itself reads x (its front kernel norms and routes), and its
                # result is published for the next layer's q|k|v by one small launch -- in_m = "moe" marks it
                cap_m, in_m, nw_m = True, "moe", None
            else:
                return
            if not (cap_a and cap_m):
                return
            plan.append((in_a, o_inv, in_m, nw_a, nw_m))
        dev = m.device
        head_perm, head_inv = ext.q_matrix_perm_info(m.lm_head.q_handle)
        norm_head = torch.empty_like(m.norm.weight)
        ext.gather_f16(m.norm.weight, head_perm, norm_head)
        self.chain = {
            "plan": plan, "head_inv": head_inv, "norm_head": norm_head,
            "xp_a": torch.zeros((self.b, cfg.hidden_size), dtype=torch.float16, device=dev),
            "xp_b": torch.zeros((self.b, cfg.hidden_size), dtype=torch.float16, device=dev),
            "ss_a": torch.zeros((self.b, 512), dtype=torch.float32, device=dev),
            "ss_b": torch.zeros((self.b, 512), dtype=torch.float32, device=dev),
        }
        # 5-16 sequences (round 4): every q_gemm launch runs on the lean kernel, whose ROWS form stages the whole rows once per
        # workgroup when rows x (K + 8) x 2 bytes fit in LDS (csrc/qgemv_lean.hip) -- launches with a larger K (down_proj) run as
        # row groups, each group streaming the weights again (the later ones from the memory-side cache).  EXL2_CHAIN_ROWGROUPS=N
        # forces groups of N rows for every launch, 0 = one group whatever the kernel underneath makes of it.
        rg = os.environ.get("EXL2_CHAIN_ROWGROUPS", "auto")
        if rg != "auto" and int(rg) > 0 and self.b > 4:
            self.chain["group_rows"] = int(rg)
        self.chain["group_auto"] = rg == "auto"
        # overlapped chain (csrc/chain_sync.h, EXPERIMENTAL): launches alternate between the decoder's stream and a second
        # one, each waits for its predecessor through words in memory.  5 launches per layer + the head (+ the gate's block).
        if os.environ.get("EXL2_CHAIN_OVERLAP", "0") != "0":
            self.chain["flags"] = torch.zeros((5 * len(plan) + 4, ext.SYNC_BLOCK_WORDS), dtype=torch.int32, device=dev)
            # (another PRIORITY class: torch hands out streams from a pool and two of them may share a hardware queue -- the launches
            # of the two graphs would then run one after the other and every wait of the first would spin until it gives up;
            # queues of different priorities are never shared)
            self.chain["stream_b"] = torch.cuda.Stream(device=dev, priority=-1) if dev.type == "cuda" else None
            if dev.type == "cuda":
                self.chain["ev_pre"], self.chain["ev_b"] = torch.cuda.Event(), torch.cuda.Event()
            self.graph_b = None
        # 5-16 sequences, EXL2_XP_TILED=1 (round 4, measured SLOWER than the default and therefore off: profiles/history/r04_xp_tiled_sweep.txt
        # -- 7B bs = 16: 6460 vs 6652 tok/s, bs = 5: 2251 vs 2565): the hand-off buffers xp in the layout the matrix cores read
        # ([K / 8][16 rows][8]: include/exl2_hip.h, exl2_chain_set_tiled) -- q|k|v, gate|up and the head then take the lean kernel's
        # XMEM form (A operands straight from memory, no staged copy); o_proj keeps the ROWS form (attention writes row-major).
        # The default keeps that layout for ONE hand-off only: gate|up -> down where down's rows do not fit the LDS (csrc/modules.hip)
        if (self.b > 4 and "flags" not in self.chain and rg == "auto" and int(os.environ.get("EXL2_LEAN_XMEM", "1")) >= 1
                and os.environ.get("EXL2_XP_TILED", "0") != "0" and os.environ.get("EXL2_LEAN", "1") != "0"):
            self.chain["xp_tiled"] = True
            for name in ("xp_a", "xp_b"):
                self.chain[name] = torch.zeros((16, cfg.hidden_size), dtype=torch.float16, device=dev)

    def step_chain(self):
        m, ext, cfg, ch = self.model, self.model.ext, self.model.config, self.chain
        b, plan = self.b, ch["plan"]
        x2 = self.x.view(b, cfg.hidden_size)
        q = m.temp_q[:b].view(b, 1, cfg.num_attention_heads, cfg.head_dim)
        k = m.temp_k[:b].view(b, 1, cfg.num_key_value_heads, cfg.head_dim)
        v = m.temp_v[:b].view(b, 1, cfg.num_key_value_heads, cfg.head_dim)
        xp_a, xp_b, ss_a, ss_b = ch["xp_a"], ch["xp_b"], ch["ss_a"], ch["ss_b"]
        overlap = "flags" in ch
        # Row groups: a launch of the lean kernel takes the rows whose staged copy fits in LDS (<= 16; <= 4 in its wave-private
        # form); more sequences run that launch once per group of rows.  A group's (xp, ss) live in its own rows of the buffers;
        # attention takes all rows in one launch.  Launches with different K may use different groupings: the number of partial
        # sums of squares per row a producer published must then be the same for all its groups (checked).
        def groups_for(K):
            g = ch.get("group_override", {}).get(K)            # what a declined launch of this K was regrouped to (_launch_step)
            if overlap or (b <= 4 and not g):
                return [(0, b)]
            g = g or ch.get("group_rows")
            if not g:
                if not ch.get("group_auto", True):
                    return [(0, b)]
                lds = int(os.environ.get("EXL2_CHAIN_ROWS_LDS", 132 * 1024))     # rows next to ~26 KB of scale rows and partial sums
                fit = lds // ((K + 8) * 2)
                # (rows that do not fit: the lean kernel's XMEM form reads the A operands from memory -- one launch for all rows;
                # EXL2_LEAN_XMEM=0: its row groups of round 4's first form)
                xmem = int(os.environ.get("EXL2_LEAN_XMEM", "1")) >= 1
                g = b if (fit >= b or xmem) else (fit if fit >= 5 else 4)
                g = -(-b // -(-b // g))                             # equal groups: 16 rows at 6 per launch -> 6 + 5 + 5
            return [(r, min(r + g, b)) for r in range(0, b, g)]

        groups_h = groups_for(cfg.hidden_size)                    # q|k|v, gate|up, head: K = hidden
        groups_o = groups_for(cfg.num_attention_heads * cfg.head_dim)
        groups_d = groups_for(cfg.intermediate_size)              # down: K = intermediate
        groups = groups_h
        pa, pb = _PartialSums(ss_a, b), _PartialSums(ss_b, b)
        pa.begin()
        tiled = bool(ch.get("xp_tiled")) and groups_h == groups_o == groups_d == [(0, b)]
        if tiled:
            ext.chain_set_tiled(True)
        try:
            for r0, r1 in groups:
                ext.embed_rows_chain(m.embed_tokens, self.ids[r0:r1], x2[r0:r1], plan[0][0], plan[0][3], xp_a[r0:r1], pa.out(r0, r1))
                pa.done(1)
        except Exception:
            ext.chain_set_tiled(False)
            raise
        if overlap:
            sa = self.stream.cuda_stream if self.stream is not None else None
            sb = ch["stream_b"].cuda_stream if ch["stream_b"] is not None else None
            ext.chain_overlap_begin(ch["flags"], sa, sb)
        try:
            # partial sums of squares per row a launch published: the count depends on which kernel took the launch (qgemv_lean:
            # one per 16-column tile, qgemv_flat: one per workgroup) and a 4-row group and a 3-row group of the same matrix may be
            # taken by different kernels -- the consumer may group the rows differently, so all groups must agree
            for i, (attn, mlp) in enumerate(m.layers):
                in_a, o_inv, in_m, nw_a, nw_m = plan[i]
                for r0, r1 in groups_h:
                    ss, cnt = pa.inp(r0, r1, "q|k|v")
                    ext.q_attn_forward_1_chain(attn.q_handle, xp_a[r0:r1], ss, cnt, r1 - r0, q[r0:r1], k[r0:r1], v[r0:r1])
                ao = attn.attend_chain(q, k, v, self.cache, self.cache_seqlens, self.block_table, o_inv)
                # every producer of the residual stream publishes it times its consumer's norm weight, in that consumer's order
                nxt, nxt_w = (plan[i + 1][0], plan[i + 1][3]) if i + 1 < len(plan) else (ch["head_inv"], ch["norm_head"])
                if in_m == "moe":
                    for r0, r1 in groups_o:
                        ext.q_attn_forward_2_chain(attn.q_handle, x2[r0:r1], ao[r0:r1], r1 - r0, None, None, None, None)
                    pa.begin()
                    if groups_h == [(0, b)]:
                        pa.done(ext.q_moe_mlp_forward_chain(mlp.q_handle, x2, b, nxt, nxt_w, xp_a, pa.out(0, b)))
                    else:
                        mlp.forward(self.x)
                        for r0, r1 in groups_h:
                            ext.publish_rows(x2[r0:r1], r1 - r0, cfg.hidden_size, nxt, nxt_w, xp_a[r0:r1], pa.out(r0, r1))
                            pa.done(1)
                    continue
                pb.begin()
                for r0, r1 in groups_o:
                    pb.done(ext.q_attn_forward_2_chain(attn.q_handle, x2[r0:r1], ao[r0:r1], r1 - r0, in_m, nw_m, xp_b[r0:r1], pb.out(r0, r1)))
                if groups_h == groups_d and os.environ.get("EXL2_CHAIN_SPLIT_MLP", "0") != "1":     # (1: test hook)
                    pa.begin()
                    for r0, r1 in groups_h:
                        ss, cnt = pb.inp(r0, r1, "gate|up")
                        pa.done(ext.q_mlp_forward_chain(mlp.q_handle, x2[r0:r1], xp_b[r0:r1], ss, cnt, r1 - r0, nxt, nxt_w,
                                                        xp_a[r0:r1], pa.out(r0, r1)))
                else:
                    # the halves separately: gate | up over its groups, then down over its own
                    for r0, r1 in groups_h:
                        ss, cnt = pb.inp(r0, r1, "gate|up")
                        ext.q_mlp_forward_chain_part(mlp.q_handle, 1, r0, x2[r0:r1], xp_b[r0:r1], ss, cnt, r1 - r0, None, None, None, None)
                    pa.begin()
                    for r0, r1 in groups_d:
                        pa.done(ext.q_mlp_forward_chain_part(mlp.q_handle, 2, r0, x2[r0:r1], None, None, 0, r1 - r0, nxt, nxt_w,
                                                             xp_a[r0:r1], pa.out(r0, r1)))
            for r0, r1 in groups_h:
                ss, cnt = pa.inp(r0, r1, "head")
                ext.gemm_half_q_half_chain(xp_a[r0:r1], ss, cnt, cfg.norm_eps, m.lm_head.q_handle, self.logits[r0:r1], r1 - r0)
        finally:
            if tiled:
                ext.chain_set_tiled(False)
            n_launches = ext.chain_overlap_end() if overlap else 0
        # greedy sampling + position increment behind the head: on the stream the head went to
        import contextlib
        on_b = overlap and ch["stream_b"] is not None and (n_launches - 1) % 2 == 1
        with (torch.cuda.stream(ch["stream_b"]) if on_b else contextlib.nullcontext()):
            self._select_token()

    def _select_token(self):
        """last launch(es) of a step: greedy arg-max, or -- while a sampled graph is being captured / run eagerly -- the device
        sampler reading its random point from the decoder's buffer, + the counter increment"""
        ext, cfg = self.model.ext, self.model.config
        sm = getattr(self, "_sampling", None)
        if sm is None:
            ext.argmax_rows(self.logits, self.ids, cfg.vocab_size, self.history, self.cache_seqlens, pos_inc=1)
            return
        ext.sample_rows_step(self.logits, sm["temperature"], sm["top_k"], sm["top_p"], sm["min_p"], sm["randoms"], sm["counter"],
                             self.ids, sm["probs"], sm["ws"], self.history, self.cache_seqlens, pos_inc=1, vocab=cfg.vocab_size)
        ext.add_i32_(sm["counter"], 1)

    def _overlapped(self) -> bool:
        return self.chain is not None and "flags" in self.chain and self.chain.get("stream_b") is not None

    def _ordered(self, launch):
        """One step of the overlapped chain = launches on two streams that are not joined inside the step.  At the step's
        boundaries each stream is put behind the other: stream B behind everything stream A was given before the step (the
        previous step's half included), stream A behind stream B's previous half; after the last step of run() stream A
        is put behind stream B once more.  (Events between graph launches, not inside them.)"""
        ch = self.chain
        ch["ev_pre"].record(self.stream)
        ch["stream_b"].wait_event(ch["ev_pre"])
        self.stream.wait_event(ch["ev_b"])
        launch()
        ch["ev_b"].record(ch["stream_b"])

    def step_eager(self):
        if self._overlapped():
            self._ordered(self._launch_step)
            if self._overlapped():
                self.stream.wait_event(self.chain["ev_b"])
            return
        self._launch_step()

    def _launch_step(self):
        while self.chain is not None:
            try:
                return self.step_chain()
            except RuntimeError as e:
                if "not covered" not in str(e):
                    raise
                # the optimistic grouping (all rows in one launch: the lean kernel's wave-private / ROWS / XMEM forms) is a guess
                # about what the kernel's host plan accepts -- XMEM is one register pass (K <= ~12 k at 4 bits), the wave-private
                # form stages rows x K-slice per wave: down_proj of a Mixtral / Llama-3 width (K = 14336) declines 5+ rows, of a
                # 70B (K = 28672) 3+.  Before giving up the WHOLE chain, regroup the launch kind that declined: row groups of
                # 4, then 2, then 1.  (a step that failed half-way is simply redone: it starts from the embedding and re-appends
                # the same cache row)
                if not self._overlapped() and self._regroup(str(e)):
                    continue
                if self._overlapped():
                    # the experimental overlapped mode never falls back silently: a run that asked for it and did not get it
                    # must not be mistaken for a measurement of it
                    raise RuntimeError("EXL2_CHAIN_OVERLAP=1: a launch of the step is outside the chained kernels (" + str(e) +
                                       "); unset EXL2_CHAIN_OVERLAP") from e
                self.chain = None               # a shape outside the chained kernels: the module-by-module route below
        m, ext, cfg = self.model, self.model.ext, self.model.config
        ext.embed_rows(m.embed_tokens, self.ids, self.x.view(self.b, cfg.hidden_size))
        for attn, mlp in m.layers:
            attn.forward(self.x, self.cache, 0, self.cache_seqlens, self.block_table)
            mlp.forward(self.x)
        ext.rms_norm(self.x.view(self.b, -1), m.norm.weight, self.xn.view(self.b, -1), cfg.norm_eps)
        ext.gemm_half_q_half(self.xn.view(self.b, -1), m.lm_head.q_handle, self.logits)
        self._select_token()

    def _regroup(self, msg: str) -> bool:
        """a chained launch was declined ("<entry point>: shape not covered"): smaller row groups for launches of its K"""
        cfg = self.model.config
        if "(down)" in msg: K = cfg.intermediate_size
        elif "q_attn_forward_2" in msg: K = cfg.num_attention_heads * cfg.head_dim
        else: K = cfg.hidden_size
        ov = self.chain.setdefault("group_override", {})
        cur = ov.get(K) or self.b
        for g in (4, 2, 1):
            if g < cur:
                ov[K] = g
                return True
        return False

    def _on_stream(self):
        import contextlib
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def capture(self):
        """Warm-up + capture run two throw-away steps; they are pointed at the LAST cache slot, whose K/V rows are
        saved and restored, so capturing never disturbs a cache that already holds a prompt."""
        ext = self.model.ext
        torch.cuda.synchronize()
        scratch_pos = self.limit - 1
        with self._on_stream():
            saved_state = (self.cache_seqlens.clone(), self.ids.clone())
            slot_tensors = list(self.cache.key_states) + list(self.cache.value_states)
            # quantized caches: the scales of the scratch slot too (they sit at the same [batch, position] index)
            slot_tensors += list(getattr(self.cache, "key_scales", [])) + list(getattr(self.cache, "value_scales", []))
            saved_kv = [t[:, scratch_pos].clone() for t in slot_tensors]
            self.cache_seqlens.fill_(scratch_pos)
            self.step_eager()                               # warm-up: lazy one-time setup must not happen under capture
            self.cache_seqlens.fill_(scratch_pos)
            self.stream.synchronize()
            if self._overlapped():
                # two graphs, one per stream, replayed side by side (the branches of ONE forked graph are run one after the
                # other on this stack: tools/probes/fork_probe.hip)
                sb = self.chain["stream_b"]
                sb.synchronize()
                ext.graph_begin_capture(self.stream.cuda_stream)
                ext.graph_begin_capture(sb.cuda_stream)
                try:
                    self._launch_step()
                finally:
                    self.graph = ext.graph_end_capture(self.stream.cuda_stream)
                    self.graph_b = ext.graph_end_capture(sb.cuda_stream)
            else:
                ext.graph_begin_capture(self.stream.cuda_stream)
                try:
                    self._launch_step()
                finally:
                    self.graph = ext.graph_end_capture(self.stream.cuda_stream)
            self.cache_seqlens.copy_(saved_state[0]); self.ids.copy_(saved_state[1])
            for t, s in zip(slot_tensors, saved_kv):
                t[:, scratch_pos].copy_(s)
            self.stream.synchronize()
        return self

    def reset(self, first_ids: torch.Tensor, seq_len: int = 0):
        with self._on_stream():
            self.ids.copy_(first_ids.to(torch.int32).view(-1))
            self.cache_seqlens.fill_(seq_len)
        self.pos = int(seq_len)

    def run(self, n_tokens: int, use_graph: bool = True):
        ext = self.model.ext
        if self.pos + n_tokens > self.limit:
            raise RuntimeError(f"decode: {self.pos} cached + {n_tokens} new tokens exceed the cache / max_seq_len ({self.limit})")
        self.pos += n_tokens
        with self._on_stream():
            sptr = self.stream.cuda_stream if self.stream is not None else None
            two = use_graph and self.graph is not None and getattr(self, "graph_b", None) is not None and self._overlapped()
            for _ in range(n_tokens):
                if two:
                    bptr = self.chain["stream_b"].cuda_stream
                    self._ordered(lambda: (ext.graph_launch(self.graph, sptr), ext.graph_launch(self.graph_b, bptr)))
                elif use_graph and self.graph is not None:
                    ext.graph_launch(self.graph, sptr)
                else:
                    self.step_eager()
            if two:
                self.stream.wait_event(self.chain["ev_b"])

    def capture_sampled(self, temperature: float, top_k: int, top_p: float = 0.0, min_p: float = 0.0, n_randoms: int = 4096):
        """A second step graph whose last launches are the device sampler (exl2_sample_rows_step) + its counter increment instead
        of the arg-max: `run_sampled` with these settings then replays ONE graph per token -- the random points are read from a
        device buffer the host fills before the run (n_randoms tokens ahead at most per fill)."""
        ext, cfg = self.model.ext, self.model.config
        if self._overlapped():
            raise RuntimeError("capture_sampled: not available on the experimental overlapped chain")
        dev = self.logits.device
        sm = {"temperature": float(temperature), "top_k": int(top_k), "top_p": float(top_p), "min_p": float(min_p),
              "randoms": torch.zeros((int(n_randoms),), dtype=torch.float32, device=dev),
              "counter": torch.zeros((1,), dtype=torch.int32, device=dev),
              "probs": torch.zeros((self.b,), dtype=torch.float32, device=dev),
              "ws": torch.empty((self.b, cfg.vocab_size), dtype=torch.float32, device=dev)}
        sm["graph"] = None
        if dev.type == "cuda":
            greedy_graph, self.graph = self.graph, None
            self._sampling = sm
            try:
                self.capture()                                   # same warm-up / scratch-slot protocol, the sampler as the last launches
                sm["graph"] = self.graph
            finally:
                self.graph = greedy_graph
                self._sampling = None
        sm["counter"].zero_()
        old = getattr(self, "_sampled", None)
        if old is not None and old.get("graph") is not None:
            ext.graph_free(old["graph"])                          # a re-capture with new settings replaces the old sampled graph
        self._sampled = sm
        return self

    def run_sampled(self, n_tokens: int, temperature: float, top_k: int, top_p: float = 0.0, min_p: float = 0.0,
                    randoms=None, seed: int = 0, use_graph: bool = True):
        """Decode `n_tokens` tokens per sequence with the DEVICE sampler (csrc/sampling.hip; the reference's sample_basic for
        these settings: ext_sampling.cpp:93-301) instead of the arg-max: per token the step's graph, then one
        `sample_rows` launch on the step's logits that overwrites the token fed to the next step, then the token log.
        The only thing the host supplies per token is the random point (`randoms[i]`, or `random.Random(seed)` like the
        reference's `random.random()` per sampling call, sampler.py:351) -- as a launch argument; nothing is read back, so
        the loop stays asynchronous like run().  Rows of a batch share the point through the reference's recurrence."""
        import random as _random
        ext, cfg = self.model.ext, self.model.config
        if self._overlapped():
            raise RuntimeError("run_sampled: not available on the experimental overlapped chain")
        if self.pos + n_tokens > self.limit:
            raise RuntimeError(f"decode: {self.pos} cached + {n_tokens} new tokens exceed the cache / max_seq_len ({self.limit})")
        if randoms is None:
            rng = _random.Random(seed)
            randoms = [rng.random() for _ in range(n_tokens)]
        if len(randoms) < n_tokens:
            raise RuntimeError("run_sampled: one random point per token is needed")
        sm = getattr(self, "_sampled", None)
        if (sm is not None and n_tokens <= sm["randoms"].numel()
                and (sm["temperature"], sm["top_k"], sm["top_p"], sm["min_p"]) == (float(temperature), int(top_k), float(top_p), float(min_p))):
            # settings of capture_sampled: the sampler is part of the step; the random points travel as ONE copy before the run,
            # then one graph launch per token (eager steps of the same launches where there is no graph)
            pts = torch.tensor([float(r) for r in randoms[:n_tokens]], dtype=torch.float32)
            self.pos += n_tokens
            with self._on_stream():
                sptr = self.stream.cuda_stream if self.stream is not None else None
                sm["randoms"][:n_tokens].copy_(pts, non_blocking=False)
                sm["counter"].zero_()
                if use_graph and sm.get("graph") is not None:
                    for _ in range(n_tokens):
                        ext.graph_launch(sm["graph"], sptr)
                else:
                    self._sampling = sm
                    try:
                        for _ in range(n_tokens):
                            self.step_eager()
                    finally:
                        self._sampling = None
            return
        if getattr(self, "_sample_ws", None) is None:
            dev = self.logits.device
            self._sample_ws = torch.empty((self.b, cfg.vocab_size), dtype=torch.float32, device=dev)
            self._sample_probs = torch.zeros((self.b,), dtype=torch.float32, device=dev)
        p0 = self.pos
        self.pos += n_tokens
        with self._on_stream():
            sptr = self.stream.cuda_stream if self.stream is not None else None
            for i in range(n_tokens):
                if use_graph and self.graph is not None:
                    ext.graph_launch(self.graph, sptr)
                else:
                    self.step_eager()
                # the step's own arg-max has written its token to ids / history[:, p0 + i + 1]; the sampled one replaces both
                ext.sample_rows(self.logits, temperature, top_k, top_p, min_p, float(randoms[i]), self.ids, self._sample_probs,
                                workspace=self._sample_ws, vocab=cfg.vocab_size)
                self.history[:, p0 + i + 1].copy_(self.ids)

    def tokens(self, start: int, n: int) -> torch.Tensor:
        """tokens generated at positions start+1 .. start+n (history[b, pos] = token sampled after `pos` cached tokens)."""
        return self.history[:, start + 1:start + 1 + n]

    def free(self):
        sm = getattr(self, "_sampled", None)
        if sm is not None and sm.get("graph") is not None:
            self.model.ext.graph_free(sm["graph"])
        self._sampled = None
        if self.graph is not None:
            self.model.ext.graph_free(self.graph)
            self.graph = None
        if getattr(self, "graph_b", None) is not None:
            self.model.ext.graph_free(self.graph_b)
            self.graph_b = None
