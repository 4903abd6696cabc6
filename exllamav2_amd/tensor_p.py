"""Tensor-parallel forward (reference tensor_p.py:102-180 split plan, linear.py:546-620 column split, attn.py:646-867 /
mlp.py:363-451 TP forwards), one process per GPU.

The reference drives all devices from ONE process: `tp_broadcast` the hidden state through pinned host memory, run the
column shards device by device, `tp_gather` the pieces back through the host (ext_tp.cpp).  Here every rank is a
process on its own GPU and holds the hidden state replicated, so the broadcast disappears and each gather is one RCCL
all-gather over xGMI (`torch.distributed`, backend nccl = RCCL; gloo in the CPU tests):

    split                     reference                                   here (rank r of N)
    q / k / v, KV cache       by KV head (`kv_split`, q = kv x groups)    KV heads [r KVH/N, (r+1) KVH/N) and their q heads
    gate / up                 128-column units of intermediate (id_split) columns [r I/N, (r+1) I/N)
    o / down                  32-column units of hidden (rs_split)        columns [r H/N, (r+1) H/N), input = gathered rows
    lm_head                   32-column units of vocab (vc_split)         columns [r V/N, (r+1) V/N)

Every linear is a COLUMN shard (`make_q_matrix_split`): an EXL2 matrix cannot be cut along K without undoing its
act-order permutation (linear.py:632-635 "Tensor not prepared for row split"), which is why the reference gathers the
activations instead of all-reducing partial sums.  Per layer: all-gather(attn_out) -> o shard -> all-gather ->
residual; all-gather(silu(gate) * up) -> down shard -> all-gather -> residual.  Each rank streams 1/N of every weight
matrix per token and owns 1/N of the KV cache; the gathered payloads are [rows, hidden | intermediate] fp16.

Column shards are contiguous copies of `q_weight[:, a:b]` / `q_scale[:, a/8:b/8]` (+ bias); q_perm / q_groups /
q_scale_max are shared by all shards of a matrix (linear.py:560-600).
"""
from __future__ import annotations

import dataclasses

import torch
import torch.distributed as dist

from .attn import ExLlamaV2Attention
import os

from .comm import all_gather, all_reduce_sum
from .linear import ExLlamaV2Linear
from .model import ExLlamaV2
from .rmsnorm import ExLlamaV2RMSNorm


def tp_split_columns(w: dict, a: int, b: int) -> dict:
    """Column shard [a, b) of an EXL2 tensor set (linear.py:560-600 `tp_split`).  `a`, `b` multiples of 32 in practice
    (q_scale packs 8 columns per word: multiples of 8 are the hard requirement, linear.py:572)."""
    if "q_weight" not in w:
        raise RuntimeError("Tensor split not implemented for GPTQ matrices")        # ext_qmatrix.cpp:113-187: EXL2 only
    if a % 8 or b % 8 or not 0 <= a < b <= w["q_weight"].shape[1]:
        raise RuntimeError(f"tp_split_columns: bad column range [{a}, {b})")
    s = {
        "q_weight": w["q_weight"][:, a:b].contiguous(),
        "q_scale": w["q_scale"][:, a // 8:b // 8].contiguous(),
        "q_scale_max": w["q_scale_max"].clone(),         # make_q_matrix_from_dict rescales it in place (ext.py:336)
        "q_groups": w["q_groups"],
        "q_invperm": w["q_invperm"],
        "q_perm": w["q_perm"],
    }
    if "bias" in w:
        s["bias"] = w["bias"][a:b].contiguous()
    return s


def _remap_4bit_columns(packed: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """int32 [rows, cols / 8], eight 4-bit values per word along the columns (q_scale): new[:, c] = old[:, index[c]]
    (tensor_remap_4bit, ext_stloader.cpp:186-219)."""
    rows, words = packed.shape
    sh = torch.arange(8, device=packed.device, dtype=torch.int64) * 4
    v = ((packed.to(torch.int64) & 0xFFFFFFFF).unsqueeze(-1) >> sh) & 0xF                 # [rows, words, 8]
    v = v.reshape(rows, words * 8)[:, index.long()].reshape(rows, words, 8)
    w = (v << sh).sum(-1)
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).contiguous()


def tp_fold_down_perm(gate_w: dict, up_w: dict, down_w: dict):
    """What ExLlamaV2MLP.load does on one device (mlp.py:162, linear.py:147-160): down_proj's act-order permutation
    is folded into the OUTPUT columns of gate / up (new column c = old column q_perm[c]), after which down_proj's packed K rows are
    simply rows 0 .. K-1 of its input -- the one EXL2 matrix of a layer that CAN be cut along K.  Returns (gate', up', down') with
    down' carrying an identity permutation.  The caller's dicts are left alone."""
    perm = down_w["q_perm"].long()
    def cols(w):
        o = dict(w)
        o["q_weight"] = w["q_weight"][:, perm].contiguous()
        o["q_scale"] = _remap_4bit_columns(w["q_scale"], perm)
        if "bias" in w: o["bias"] = w["bias"][perm].contiguous()
        return o
    d = dict(down_w)
    ident = torch.arange(perm.shape[0], device=perm.device, dtype=down_w["q_perm"].dtype)
    d["q_perm"], d["q_invperm"] = ident, ident.clone()
    return cols(gate_w), cols(up_w), d


def tp_split_rows(w: dict, ka: int, kb: int):
    """Packed K rows [ka, kb) of an EXL2 tensor set whose permutation is the identity (tp_fold_down_perm): the groups that cover
    exactly that range -- their words of q_weight, their rows of q_scale / q_scale_max, q_groups rebased.  None when the range does
    not fall on group boundaries (the caller keeps the column split)."""
    qg = w["q_groups"].cpu().tolist()
    groups = len(qg) // 2
    total_words = w["q_weight"].shape[0]
    k, ga, gb, wa, wb = 0, None, None, None, None
    for g in range(groups):
        bits, off = qg[2 * g], qg[2 * g + 1]
        nxt = qg[2 * g + 3] if g + 1 < groups else total_words
        rows = (nxt - off) * 32 // bits
        if k == ka: ga, wa = g, off
        k += rows
        if k == kb: gb, wb = g + 1, nxt
    if ga is None or gb is None or gb <= ga:
        return None
    new_groups = []
    for g in range(ga, gb):
        new_groups += [qg[2 * g], qg[2 * g + 1] - wa]
    n = kb - ka
    ident = torch.arange(n, device=w["q_weight"].device, dtype=w["q_perm"].dtype)
    return {
        "q_weight": w["q_weight"][wa:wb].contiguous(),
        "q_scale": w["q_scale"][ga:gb].contiguous(),
        "q_scale_max": w["q_scale_max"][ga:gb].clone(),
        "q_groups": torch.tensor(new_groups, dtype=w["q_groups"].dtype, device=w["q_groups"].device),
        "q_perm": ident, "q_invperm": ident.clone(),
    }                                                       # (bias: added once, by the caller, after the reduction)


def tp_ranges(total: int, world: int, unit: int):
    """Equal column ranges in units of `unit` columns (tensor_p.py:118-180 builds uneven splits over devices with
    different free memory; ranks here are identical GPUs and all-gather wants equal pieces)."""
    if total % (world * unit):
        raise RuntimeError(f"tensor-parallel split: {total} columns do not divide into {world} x {unit}-column units")
    n = total // world
    return [(r * n, (r + 1) * n) for r in range(world)]


class TPContext:
    """What the reference keeps in `TPContext` / `ExtTPContext` (tensor_p.py:20-100, ext_tp.h): the split plan and the
    communication buffers.  One per rank."""

    def __init__(self, cfg, rank: int, world: int, device, max_rows: int, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.device = torch.device(device)
        h, inter = cfg.hidden_size, cfg.intermediate_size
        if cfg.num_key_value_heads % world:
            raise RuntimeError(f"tensor-parallel split: {cfg.num_key_value_heads} KV heads over {world} ranks")
        g = cfg.num_attention_heads // cfg.num_key_value_heads
        kvl = cfg.num_key_value_heads // world
        self.kv_split = (rank * kvl, (rank + 1) * kvl)                              # KV heads of this rank
        self.q_split = (rank * kvl * g, (rank + 1) * kvl * g)                       # their query heads
        self.id_split = tp_ranges(inter, world, 32)[rank]
        self.rs_split = tp_ranges(h, world, 32)[rank]
        vpad = (cfg.vocab_size + 31) // 32 * 32
        self.vocab_padded = (vpad + 32 * world - 1) // (32 * world) * (32 * world)  # pad so the vocab splits evenly
        self.vc_split = tp_ranges(self.vocab_padded, world, 32)[rank]
        self.max_rows = max_rows
        widest = max(h, inter, cfg.num_attention_heads * cfg.head_dim, self.vocab_padded)
        self._gather = torch.empty((world * max_rows * (widest // world),), dtype=torch.float16, device=self.device)
        self._full = torch.empty((max_rows * widest,), dtype=torch.float16, device=self.device)

    def all_gather_columns(self, local: torch.Tensor) -> torch.Tensor:
        """[rows, n/N] per rank -> [rows, n] with rank r's columns at [r n/N, (r+1) n/N) (tp_gather, ext_tp.cpp).
        One all-gather; for one row the gathered buffer already IS the result."""
        rows, nl = local.shape
        w = self.world
        if w == 1:
            return local
        buf = self._gather[:w * rows * nl].view(w, rows, nl)
        all_gather(list(buf.unbind(0)), local.contiguous(), group=self.group)
        if rows == 1:
            return buf.view(1, w * nl)
        full = self._full[:rows * w * nl].view(rows, w, nl)
        full.copy_(buf.transpose(0, 1))
        return full.view(rows, w * nl)


class TPLinear(ExLlamaV2Linear):
    """Column shard of a linear (linear.py:546-620): forward = local q_gemm on the shard; the caller gathers."""

    def __init__(self, ext, key, in_features, out_features_full, col_range, model=None):
        a, b = col_range
        super().__init__(ext, key, in_features, b - a, model)
        self.col_range = col_range
        self.out_features_full = out_features_full

    def load(self, w_full: dict, temp_dq=None, prescale: float = 1.0):
        from .ext import none_tensor
        self.q_tensors = tp_split_columns(w_full, *self.col_range)
        self.q_handle = self.ext.make_q_matrix_from_dict(self.q_tensors, temp_dq if temp_dq is not None else none_tensor,
                                                         key=self.key, prescale=prescale, split=True)
        return self


class TPAttention(ExLlamaV2Attention):
    """attn.py:646-867 (`forward_tp`): local heads' q/k/v + attention over the local slice of the KV cache, gather,
    column shard of o_proj, gather, residual."""

    def __init__(self, model, key: str, layer_idx: int):
        ctx, full = model.tp, model.full_config
        self.model, self.ext, self.key, self.layer_idx = model, model.ext, key, layer_idx
        h, hd = full.hidden_size, full.head_dim
        q0, q1 = ctx.q_split
        k0, k1 = ctx.kv_split
        self.q_proj = TPLinear(self.ext, key + ".self_attn.q_proj", h, full.num_attention_heads * hd, (q0 * hd, q1 * hd), model)
        self.k_proj = TPLinear(self.ext, key + ".self_attn.k_proj", h, full.num_key_value_heads * hd, (k0 * hd, k1 * hd), model)
        self.v_proj = TPLinear(self.ext, key + ".self_attn.v_proj", h, full.num_key_value_heads * hd, (k0 * hd, k1 * hd), model)
        self.o_proj = TPLinear(self.ext, key + ".self_attn.o_proj", full.num_attention_heads * hd, h, ctx.rs_split, model)
        self.pre_layernorm = None
        self.q_handle = None
        self.fused_decode, self.q4_fused = True, True

    # load() and _project_qkv() are the single-device ones: the fused front half (one launch: RMSNorm + q | k | v)
    # runs on the shards with the LOCAL head counts of model.config; the handle's o_proj slot (the o shard) is only
    # used by q_attn_forward_2, which this class replaces:

    def _project_out(self, hidden_states, attn_out, b, q_len):
        rows = b * q_len
        tp = self.model.tp
        full = tp.all_gather_columns(attn_out.view(rows, -1))                  # [rows, H * hd], heads in rank order
        o = tp.all_gather_columns(self.o_proj.forward(full))
        hidden_states.view(rows, -1).add_(o)
        return hidden_states


class TPMLP:
    """mlp.py:363-451 (`forward_tp`): gate / up column shards -> act * mul -> down.  Two forms of the second half:

    * row-parallel down_proj (round 6; default where the tensors allow it): down_proj's act-order permutation is folded into the
      columns of gate / up at load (what mlp.py:162 / linear.py:147-160 do on one device), gate / up are cut along those PERMUTED
      columns, down_proj along its packed K rows [r I/N, (r+1) I/N) -- every rank multiplies ITS slice of silu(gate) * up by ITS rows
      of down_proj and ONE all-reduce of [rows, hidden] partial sums (fp32 on the wire) replaces the all-gather of the intermediate
      rows AND the all-gather of the outputs: three collectives per layer instead of four, and the 2 x intermediate / N values per
      row never travel.  (o_proj cannot take this: its permutation mixes the heads.)
    * column shards + two all-gathers (the reference's plan; kept for GPTQ tensors, a range that does not fall on group
      boundaries, EXL2_TP_ROW_DOWN=0)."""

    def __init__(self, model, key: str, layer_idx: int):
        ctx, full = model.tp, model.full_config
        self.model, self.ext, self.key, self.layer_idx = model, model.ext, key, layer_idx
        h, inter = full.hidden_size, full.intermediate_size
        self.gate_proj = TPLinear(self.ext, key + ".mlp.gate_proj", h, inter, ctx.id_split, model)
        self.up_proj = TPLinear(self.ext, key + ".mlp.up_proj", h, inter, ctx.id_split, model)
        self.down_proj = TPLinear(self.ext, key + ".mlp.down_proj", inter, h, ctx.rs_split, model)
        self.post_attention_layernorm = None
        self.row_down = False
        self.down_bias = None

    def load(self, ck: dict):
        from .ext import none_tensor
        ctx = self.model.tp
        gw, uw, dw = ck[self.gate_proj.key], ck[self.up_proj.key], ck[self.down_proj.key]
        rows = None
        if ctx.world > 1 and os.environ.get("EXL2_TP_ROW_DOWN", "1") != "0" and all("q_weight" in w for w in (gw, uw, dw)):
            g2, u2, d2 = tp_fold_down_perm(gw, uw, dw)
            rows = tp_split_rows(d2, *ctx.id_split)
        if rows is not None:
            self.row_down = True
            self.gate_proj.load(g2); self.up_proj.load(u2)                       # column shards of the permuted columns
            a, b = ctx.id_split
            lin = self.down_proj
            lin.in_features, lin.out_features, lin.padding = b - a, self.model.full_config.hidden_size, 0
            lin.col_range = (0, lin.out_features)
            lin.q_tensors = rows
            lin.q_handle = self.ext.make_q_matrix_from_dict(rows, none_tensor, key=lin.key)
            self.down_bias = dw.get("bias")
        else:
            for lin in (self.gate_proj, self.up_proj, self.down_proj):
                lin.load(ck[lin.key])
        self.post_attention_layernorm = ck[self.key + ".post_attention_layernorm"]
        return self

    def unload(self):
        for lin in (self.gate_proj, self.up_proj, self.down_proj):
            lin.unload()

    def forward(self, hidden_states: torch.Tensor):
        m, ext, tp = self.model, self.ext, self.model.tp
        h = hidden_states.shape[-1]
        rows = hidden_states.numel() // h
        x2 = hidden_states.view(rows, h)
        xn = m.temp_state[:rows]
        ext.rms_norm(x2, self.post_attention_layernorm, xn, m.config.norm_eps)
        g = self.gate_proj.forward(xn)
        u = self.up_proj.forward(xn)
        ext.act_mul_(g, u)                                                       # silu(g) * u in place on g
        if self.row_down:
            part = self.down_proj.forward(g).float()                             # this rank's K rows: partial sums [rows, hidden]
            all_reduce_sum(part, group=tp.group)
            if self.down_bias is not None: part += self.down_bias.float()
            x2.add_(part.half())
            return hidden_states
        a = tp.all_gather_columns(g)
        x2.add_(tp.all_gather_columns(self.down_proj.forward(a)))
        return hidden_states


class TPHead:
    """lm_head column shard + gather (vc_split); `forward` returns the full (padded) logits on every rank."""

    def __init__(self, model, w_full: dict):
        ctx, full = model.tp, model.full_config
        vpad = w_full["q_weight"].shape[1]
        if vpad != ctx.vocab_padded:
            raise RuntimeError(f"lm_head has {vpad} columns; this split needs {ctx.vocab_padded} (pad the vocab to 32 x ranks)")
        self.model = model
        self.shard = TPLinear(model.ext, "lm_head", full.hidden_size, vpad, ctx.vc_split, model).load(w_full)
        self.q_handle = self.shard.q_handle

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        rows = x.numel() // x.shape[-1]
        out = self.model.tp.all_gather_columns(self.shard.forward(x.view(rows, -1)))
        return out.view(x.shape[:-1] + (out.shape[-1],))

    def unload(self):
        self.shard.unload()

    def weight_bytes(self) -> int:
        return self.shard.weight_bytes()


class ExLlamaV2TP(ExLlamaV2):
    """One rank of a tensor-parallel model.  `self.config` carries the LOCAL head / intermediate counts (scratch, KV
    cache and attention kernels are sized by it), `self.full_config` the model's.  forward / forward_chunk are the
    single-device ones: the modules hide the collectives."""

    def __init__(self, config, rank: int, world: int, device="cuda:0", ext=None, group=None):
        if getattr(config, "num_experts", 0):
            raise RuntimeError("tensor parallel: MoE blocks are not split (architecture.py:291-305 has no supports_tp)")
        max_rows = config.max_input_len * max(1, config.max_batch_size)
        self.full_config = config
        self.tp = TPContext(config, rank, world, device, max_rows, group)
        local = dataclasses.replace(
            config,
            num_attention_heads=config.num_attention_heads // world,
            num_key_value_heads=config.num_key_value_heads // world,
            intermediate_size=config.intermediate_size // world)
        super().__init__(local, device=device, ext=ext)
        # the prefill policy route (reconstruct + library GEMM) works per shard as it does per matrix

    def load(self, ck: dict, layers=None):
        """`ck` holds FULL tensors (every rank reads the same checkpoint, like the reference's loader); each module
        keeps only its column shards."""
        cfg = self.full_config
        self.layer_ids = list(range(cfg.num_hidden_layers) if layers is None else layers)
        for local_idx, i in enumerate(self.layer_ids):
            key = f"model.layers.{i}"
            attn = TPAttention(self, key, local_idx).load(ck)
            mlp = TPMLP(self, key, local_idx).load(ck)
            self.layers.append((attn, mlp))
            self.modules += [attn, mlp]
        self._load_ends(ck)
        return self

    def load_more(self, ck: dict, layer: int):
        """Incremental load: one layer's full tensors at a time, so a rank never holds more than one unsharded layer
        (70B at 2.5 bpw: 0.27 GB per layer instead of 22 GB)."""
        key = f"model.layers.{layer}"
        if not hasattr(self, "layer_ids"): self.layer_ids = []
        local_idx = len(self.layers)
        attn = TPAttention(self, key, local_idx).load(ck)
        mlp = TPMLP(self, key, local_idx).load(ck)
        self.layers.append((attn, mlp))
        self.modules += [attn, mlp]
        self.layer_ids.append(layer)
        self._load_ends(ck)
        return self

    def _load_ends(self, ck: dict):
        cfg = self.full_config
        if "model.embed_tokens" in ck:
            self.embed_tokens = ck["model.embed_tokens"]
        self.vocab_padded = self.tp.vocab_padded
        if "lm_head" in ck:
            self.norm = ExLlamaV2RMSNorm(self.ext, "model.norm", ck["model.norm"], cfg.norm_eps)
            self.lm_head = TPHead(self, ck["lm_head"])
        self.loaded = True


class TPGreedyDecoder:
    """test_inference.py:604-609 (forward(ids[:, -1:]); argmax; append) on a tensor-parallel model: every rank runs the
    same step on its shards and ends up with the same token (the logits are gathered, argmax is deterministic), so no
    token broadcast is needed.  Positions and token feedback live on the device, like GreedyGraphDecoder."""

    def __init__(self, model: ExLlamaV2TP, cache, batch_size: int = 1):
        from .cache import PAGE_SIZE
        self.model, self.cache, self.b = model, cache, batch_size
        cfg, dev = model.full_config, model.device
        pages = cache.max_seq_len // PAGE_SIZE
        assert cache.max_seq_len % PAGE_SIZE == 0
        self.block_table = torch.arange(batch_size * pages, dtype=torch.int32, device=dev).view(batch_size, pages).contiguous()
        self.cache_seqlens = torch.zeros((batch_size,), dtype=torch.int32, device=dev)
        self.ids = torch.zeros((batch_size,), dtype=torch.int32, device=dev)
        self.history = torch.zeros((batch_size, cache.max_seq_len + 2), dtype=torch.int32, device=dev)
        self.x = torch.zeros((batch_size, 1, cfg.hidden_size), dtype=torch.float16, device=dev)
        self.xn = torch.zeros_like(self.x)
        self.pos, self.limit = 0, min(cache.max_seq_len, cfg.max_seq_len)

    def reset(self, first_ids: torch.Tensor, seq_len: int = 0):
        self.ids.copy_(first_ids.to(torch.int32).view(-1))
        self.cache_seqlens.fill_(seq_len)
        self.pos = int(seq_len)

    def step(self):
        m, ext, cfg = self.model, self.model.ext, self.model.full_config
        ext.embed_rows(m.embed_tokens, self.ids, self.x.view(self.b, cfg.hidden_size))
        for attn, mlp in m.layers:
            attn.forward(self.x, self.cache, 0, self.cache_seqlens, self.block_table)
            mlp.forward(self.x)
        ext.rms_norm(self.x.view(self.b, -1), m.norm.weight, self.xn.view(self.b, -1), cfg.norm_eps)
        logits = m.lm_head.forward(self.xn.view(self.b, -1)).contiguous()
        ext.add_i32_(self.cache_seqlens, 1)
        ext.argmax_rows(logits, self.ids, cfg.vocab_size, self.history, self.cache_seqlens)

    def run(self, n_tokens: int):
        if self.pos + n_tokens > self.limit:
            raise RuntimeError(f"decode: {self.pos} cached + {n_tokens} new tokens exceed the cache / max_seq_len ({self.limit})")
        self.pos += n_tokens
        for _ in range(n_tokens):
            self.step()

    def tokens(self, start: int, n: int) -> torch.Tensor:
        return self.history[:, start + 1:start + 1 + n]


def run_tp_bench(cfg, args, rank: int, world: int, device, ext=None):
    """bench.py backend for `--gpus N --parallel tp`: ONE sequence (per batch row) decoded by all ranks together --
    strong scaling (total work per token fixed, 1/N of the weight bytes per rank).  Eager launches: the step contains
    collectives, so it is not captured into a HIP graph here."""
    import time

    from .cache import ExLlamaV2Cache, ExLlamaV2Cache_Q4
    from .synth import synth_checkpoint
    max_seq = max(2048, ((args.ctx + args.steps + args.warmup + 1 + 255) // 256) * 256)
    cfg = dataclasses.replace(cfg, max_seq_len=max_seq)
    t_load = time.perf_counter()
    sync = torch.cuda.synchronize if torch.device(device).type == "cuda" else (lambda: None)
    model = ExLlamaV2TP(cfg, rank, world, device=device, ext=ext)
    # every rank draws the same full tensors (same seed) layer by layer and keeps only its shards
    for i in range(cfg.num_hidden_layers):
        ck = synth_checkpoint(cfg, device, recipe=args.recipe, seed=0, layers=[i], with_embed=(i == 0),
                              with_head=(i == cfg.num_hidden_layers - 1))
        _pad_head(ck, model.tp.vocab_padded)
        model.load_more(ck, i)
        del ck
    sync()
    t_load = time.perf_counter() - t_load
    cache_cls = ExLlamaV2Cache_Q4 if getattr(args, "cache", "fp16") == "q4" else ExLlamaV2Cache
    cache = cache_cls(model, batch_size=args.batch, max_seq_len=max_seq)
    dec = TPGreedyDecoder(model, cache, batch_size=args.batch)
    dec.reset(torch.tensor([1] * args.batch), args.ctx)
    dec.run(args.warmup)
    sync()
    dist.barrier()
    t0 = time.perf_counter()
    dec.run(args.steps)
    sync()
    dist.barrier()
    tdev = "cpu" if dist.get_backend() == "gloo" else device      # (bookkeeping scalars: host tensors on gloo)
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=tdev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    toks = dec.tokens(args.ctx + args.warmup, args.steps)
    assert int(toks.min()) >= 0 and int(toks.max()) < cfg.vocab_size
    wb = torch.tensor([float(model.weight_bytes())], dtype=torch.float64, device=tdev)
    gathered = [torch.zeros_like(wb) for _ in range(world)]
    dist.all_gather(gathered, wb)
    return {"value": args.batch * args.steps / dt, "ms_per_step": dt / args.steps * 1e3, "load_s": t_load,
            "scaling": "strong", "parallelism": f"tensor parallel x{world} (column shards + all-gather over RCCL), eager launches",
            "weight_bytes_per_rank": [int(g.item()) for g in gathered]}


def _pad_head(ck: dict, vocab_padded: int):
    """Zero-extend lm_head's columns to the evenly splitting width (padded logits columns are never sampled: argmax
    looks at vocab_size columns)."""
    w = ck.get("lm_head")
    if w is None or w["q_weight"].shape[1] == vocab_padded:
        return
    extra = vocab_padded - w["q_weight"].shape[1]
    w["q_weight"] = torch.nn.functional.pad(w["q_weight"], (0, extra)).contiguous()
    w["q_scale"] = torch.nn.functional.pad(w["q_scale"], (0, extra // 8)).contiguous()
