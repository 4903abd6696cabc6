"""Layer-split pipeline across ranks (world_size 2, gloo, CPU emulation backend): tokens must equal the single-process
model's greedy tokens exactly (same kernels, deterministic reductions)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import ROOT, build_emu_if_needed


def _cfg():
    from exllamav2_amd.config import ExLlamaV2Config
    return ExLlamaV2Config(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                           num_key_value_heads=1, head_dim=64, vocab_size=96, max_seq_len=256, max_input_len=32)


def _emu_ext():
    from exllamav2_amd import _lib
    from exllamav2_amd.ext import ExtC
    return ExtC(_lib.Lib(build_emu_if_needed()), allow_cpu=True)


def _worker(rank, world, port, n_ticks, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from exllamav2_amd.pipeline import PipelineStage, run_pipeline
    stage = PipelineStage(_cfg(), rank, world, "cpu", n_seqs=world, max_seq_len=256, seed=5, ext=_emu_ext(), use_graph=False)
    assert stage.chain is not None                      # dense chain-capable layers: the stage runs the chained decode route
    sampled = run_pipeline(stage, [3, 11], n_ticks)
    assert stage.chain is not None                      # ... and did not fall back on the way
    if rank == world - 1:
        np.save(out_path, stage.history.numpy())
        assert sampled == n_ticks - (world - 1)
    dist.barrier()
    dist.destroy_process_group()


def test_layer_split_pipeline_matches_single_process(tmp_path):
    world, n_ticks = 2, 7                      # rank 1 samples at ticks 1..6: 3 tokens for each of the 2 sequences
    out = str(tmp_path / "hist.npy")
    build_emu_if_needed()
    mp.spawn(_worker, args=(world, 29533, n_ticks, out), nprocs=world, join=True)
    hist = np.load(out)
    # single-process reference with the same checkpoint slices
    from exllamav2_amd.cache import ExLlamaV2Cache
    from exllamav2_amd.model import ExLlamaV2, GreedyGraphDecoder
    from exllamav2_amd.synth import synth_checkpoint
    cfg = _cfg()
    model = ExLlamaV2(cfg, device="cpu", ext=_emu_ext()).load(synth_checkpoint(cfg, "cpu", seed=5))
    for s, tok0 in enumerate([3, 11]):
        cache = ExLlamaV2Cache(model, batch_size=1, max_seq_len=256)
        dec = GreedyGraphDecoder(model, cache, batch_size=1)
        dec.reset(torch.tensor([tok0]), 0)
        dec.run(3, use_graph=False)
        want = dec.tokens(0, 3).numpy()[0]
        assert np.array_equal(hist[s, 1:4], want), (s, hist[s, :5], want)
    model.unload()


def _worker_depth2(rank, world, port, n_ticks, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from exllamav2_amd.pipeline import PipelineStage, run_pipeline
    stage = PipelineStage(_cfg(), rank, world, "cpu", n_seqs=2 * world, max_seq_len=256, seed=5, ext=_emu_ext(), use_graph=False,
                          depth=2)
    sampled = run_pipeline(stage, [3, 11, 40, 77], n_ticks)
    if rank == world - 1:
        np.save(out_path, stage.history.numpy())
        assert sampled == n_ticks - 2 * (world - 1)
    dist.barrier()
    dist.destroy_process_group()


def test_layer_split_pipeline_double_buffered_schedule(tmp_path):
    """depth 2 (the overlapped hand-off's schedule: 2 N sequences in flight, sequence s at rank r at ticks s + 2 r mod 2 N,
    message buffer pair = tick parity): every sequence's tokens equal the single-process decoder's.  (gloo executes the
    exchanges on the host thread; the stream / event ordering of the GPU form is exercised on hardware.)"""
    world, n_ticks = 2, 14                     # rank 1 samples at ticks 2..13: 3 tokens for each of the 4 sequences
    out = str(tmp_path / "hist.npy")
    build_emu_if_needed()
    mp.spawn(_worker_depth2, args=(world, 29541, n_ticks, out), nprocs=world, join=True)
    hist = np.load(out)
    from exllamav2_amd.cache import ExLlamaV2Cache
    from exllamav2_amd.model import ExLlamaV2, GreedyGraphDecoder
    from exllamav2_amd.synth import synth_checkpoint
    cfg = _cfg()
    model = ExLlamaV2(cfg, device="cpu", ext=_emu_ext()).load(synth_checkpoint(cfg, "cpu", seed=5))
    for s, tok0 in enumerate([3, 11, 40, 77]):
        cache = ExLlamaV2Cache(model, batch_size=1, max_seq_len=256)
        dec = GreedyGraphDecoder(model, cache, batch_size=1)
        dec.reset(torch.tensor([tok0]), 0)
        dec.run(3, use_graph=False)
        want = dec.tokens(0, 3).numpy()[0]
        assert np.array_equal(hist[s, 1:4], want), (s, hist[s, :5], want)
    model.unload()


def _bench_worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import argparse
    from exllamav2_amd.pipeline import run_layer_split_bench
    args = argparse.Namespace(ctx=3, steps=4, warmup=2, ramp=2, recipe="4.0bpw", no_graph=True, cache="fp16")
    r = run_layer_split_bench(_cfg(), args, rank, world, "cpu", ext=_emu_ext())
    assert r["value"] > 0 and r["ms_per_step"] > 0
    # the self-check fields of a driver-run scaling line: one weight figure per rank (the last rank also holds the head)
    wb = r["weight_bytes_per_rank"]
    assert len(wb) == world and all(b > 0 for b in wb) and wb[-1] > wb[0]
    assert f"ranks: {world}" in r["parallelism"]
    assert r["sequences_in_flight"] == 2 * world and len(r["per_gpu_weight_roofline_frac"]) == world
    dist.destroy_process_group()


def test_layer_split_bench_backend_runs_on_gloo():
    """The function `bench.py --gpus N` calls for N > 1 (pipe fill, warm-up, timed ticks, max-over-ranks timing), end to end."""
    build_emu_if_needed()
    mp.spawn(_bench_worker, args=(2, 29535), nprocs=2, join=True)


def _moe_cfg():
    from exllamav2_amd.config import ExLlamaV2Config
    return ExLlamaV2Config(hidden_size=128, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                           num_key_value_heads=1, head_dim=64, vocab_size=96, max_seq_len=256, max_input_len=32,
                           num_experts=4, num_experts_per_token=2, arch="mixtral")


def _moe_worker(rank, world, port, n_ticks, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from exllamav2_amd.pipeline import PipelineStage, run_pipeline
    stage = PipelineStage(_moe_cfg(), rank, world, "cpu", n_seqs=world, max_seq_len=256, recipe="3.5bpw", seed=6,
                          ext=_emu_ext(), use_graph=False)
    run_pipeline(stage, [4, 9], n_ticks)
    if rank == world - 1:
        np.save(out_path, stage.history.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_layer_split_pipeline_with_moe_blocks(tmp_path):
    """BASELINE configs[4] in miniature: a Mixtral-style model (sparse MoE MLP, 3.5 bpw mix) over the layer split --
    the reference's own multi-GPU mode for MoE (no tensor parallel for Mixtral)."""
    world, n_ticks = 2, 5
    out = str(tmp_path / "hist.npy")
    build_emu_if_needed()
    mp.spawn(_moe_worker, args=(world, 29537, n_ticks, out), nprocs=world, join=True)
    hist = np.load(out)
    from exllamav2_amd.cache import ExLlamaV2Cache
    from exllamav2_amd.model import ExLlamaV2, GreedyGraphDecoder
    from exllamav2_amd.synth import synth_checkpoint
    cfg = _moe_cfg()
    model = ExLlamaV2(cfg, device="cpu", ext=_emu_ext()).load(synth_checkpoint(cfg, "cpu", recipe="3.5bpw", seed=6))
    for s, tok0 in enumerate([4, 9]):
        cache = ExLlamaV2Cache(model, batch_size=1, max_seq_len=256)
        dec = GreedyGraphDecoder(model, cache, batch_size=1)
        dec.reset(torch.tensor([tok0]), 0)
        dec.run(2, use_graph=False)
        assert np.array_equal(hist[s, 1:3], dec.tokens(0, 2).numpy()[0]), (s, hist[s, :4])
    model.unload()


def test_split_layers_partition():
    from exllamav2_amd.pipeline import split_layers
    for L in (2, 22, 32, 80):
        for w in (1, 2, 4, 8):
            parts = [split_layers(L, w, r) for r in range(w)]
            assert sum(parts, []) == list(range(L))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


@pytest.mark.gpu
def test_pipeline_stage_graph_on_gpu():
    """The per-sequence HIP graphs of a pipeline stage (world = 1: the stage is first and last), fed back by hand."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from exllamav2_amd.cache import ExLlamaV2Cache
    from exllamav2_amd.model import ExLlamaV2, GreedyGraphDecoder
    from exllamav2_amd.pipeline import PipelineStage
    from exllamav2_amd.synth import synth_checkpoint
    cfg = _cfg()
    stage = PipelineStage(cfg, 0, 1, "cuda:0", n_seqs=2, max_seq_len=256, seed=5).capture()
    toks = {0: [], 1: []}
    cur = {0: 3, 1: 11}
    for _ in range(3):
        for s in (0, 1):
            with stage._on_stream():
                stage.msg_in.zero_()
                stage.msg_in[:2].view(torch.int32).copy_(torch.tensor([cur[s]], dtype=torch.int32).cuda())
            stage.step(s)
            torch.cuda.synchronize()
            cur[s] = int(stage.msg_out[:2].view(torch.int32).item())
            toks[s].append(cur[s])
    model = ExLlamaV2(cfg, device="cuda:0").load(synth_checkpoint(cfg, "cuda:0", seed=5))
    for s, tok0 in enumerate([3, 11]):
        cache = ExLlamaV2Cache(model, batch_size=1, max_seq_len=256)
        dec = GreedyGraphDecoder(model, cache, batch_size=1).capture()
        dec.reset(torch.tensor([tok0]), 0)
        dec.run(3)
        torch.cuda.synchronize()
        assert dec.tokens(0, 3).cpu().numpy()[0].tolist() == toks[s]
        dec.free()
    stage.free()


def _stages_by_hand(be, world, chain, n_tokens, firsts, use_graph):
    """All `world` stages of a layer split in ONE process on the backend's device, messages passed by hand (what RCCL / gloo
    do between ranks): first, middle and last stages, chained decode route on or off."""
    from exllamav2_amd.config import ExLlamaV2Config
    from exllamav2_amd.pipeline import PipelineStage
    cfg = ExLlamaV2Config(hidden_size=128, intermediate_size=256, num_hidden_layers=world + 1, num_attention_heads=2,
                          num_key_value_heads=1, head_dim=64, vocab_size=96, max_seq_len=256, max_input_len=32)
    old = os.environ.get("EXL2_CHAIN")
    os.environ["EXL2_CHAIN"] = "1" if chain else "0"
    try:
        stages = [PipelineStage(cfg, r, world, be.device, n_seqs=len(firsts), max_seq_len=256, seed=5, ext=be.ext,
                                use_graph=use_graph) for r in range(world)]
    finally:
        if old is None: os.environ.pop("EXL2_CHAIN")
        else: os.environ["EXL2_CHAIN"] = old
    assert all((st.chain is not None) == chain for st in stages)
    for st in stages:
        st.capture()
    out = {s: [] for s in range(len(firsts))}
    cur = dict(enumerate(firsts))
    for _ in range(n_tokens):
        for s in range(len(firsts)):
            msg = torch.zeros_like(stages[0].msg_in)
            msg[:2].view(torch.int32).copy_(torch.tensor([cur[s]], dtype=torch.int32))
            for st in stages:
                with st._on_stream():
                    st.msg_in.copy_(msg)
                st.step(s)
                if be.device != "cpu":
                    torch.cuda.synchronize()
                msg = st.msg_out.clone()
            cur[s] = int(msg[:2].view(torch.int32).item())
            out[s].append(cur[s])
    for st in stages:
        st.free()
    return cfg, out


def test_pipeline_first_middle_last_stage_chained_equals_unchained_equals_single_process(be):
    """Three stages over four layers (so one stage holds two layers): the chained decode route inside a stage (what
    `bench.py --gpus N` runs) against the module-by-module route and against the single-process decoder, token for token."""
    from exllamav2_amd.cache import ExLlamaV2Cache
    from exllamav2_amd.model import ExLlamaV2, GreedyGraphDecoder
    from exllamav2_amd.synth import synth_checkpoint
    graph = be.device != "cpu"
    cfg, chained = _stages_by_hand(be, 3, True, 4, [3, 11], graph)
    _, plain = _stages_by_hand(be, 3, False, 4, [3, 11], graph)
    assert chained == plain
    model = ExLlamaV2(cfg, device=be.device, ext=be.ext).load(synth_checkpoint(cfg, be.device, seed=5))
    for s, tok0 in enumerate([3, 11]):
        cache = ExLlamaV2Cache(model, batch_size=1, max_seq_len=256)
        dec = GreedyGraphDecoder(model, cache, batch_size=1)
        dec.reset(torch.tensor([tok0]), 0)
        dec.run(4, use_graph=False)
        assert be.n(dec.tokens(0, 4))[0].tolist() == chained[s], (s, chained[s])
    model.unload()
