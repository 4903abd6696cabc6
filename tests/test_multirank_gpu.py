"""The multi-GPU schedules on the REAL library with world_size 2 on a ONE-GPU box: both ranks drive cuda:0 (RCCL refuses two ranks
on one device, so the wire is gloo with host-staged messages -- exllamav2_amd/comm.py); everything else is what an N-GPU run
executes: libexl2_hip.so kernels, per-stage HIP graphs, the depth-2 double-buffered hand-off on a second stream with its events,
column shards through make_q_matrix_split, the gathered logits.  Checker: the single-process decoder on the same GPU + the oracle.

Plus the `bench.py --gpus N` launcher contract (never an n_gpus = 1 line for --gpus 2)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import ROOT

LOGIT_TOL = 0.03
PROMPT = [3, 17, 42, 7, 99]
N_DECODE = 3


def _cfg():
    from exllamav2_amd.config import ExLlamaV2Config
    return ExLlamaV2Config(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                           num_key_value_heads=2, head_dim=64, vocab_size=120, max_seq_len=256, max_input_len=32)


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)


def _pipeline_worker(rank, world, port, n_ticks, out_path):
    _init(rank, world, port)
    from exllamav2_amd.pipeline import PipelineStage, run_pipeline
    stage = PipelineStage(_cfg(), rank, world, "cuda:0", n_seqs=2 * world, max_seq_len=256, seed=5, use_graph=True, depth=2)
    assert stage.chain is not None and stage.comm_stream is not None
    stage.capture()
    assert len(stage.graphs) == 2 * world                    # one HIP graph per sequence in flight
    sampled = run_pipeline(stage, [3, 11, 40, 77], n_ticks)
    torch.cuda.synchronize()
    assert stage.chain is not None
    if rank == world - 1:
        np.save(out_path, stage.history.cpu().numpy())
        assert sampled == n_ticks - 2 * (world - 1)
    stage.free()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_layer_split_pipeline_two_ranks_real_library(tmp_path):
    """depth-2 layer-split pipeline, graphs + second-stream hand-off, two ranks on cuda:0: every sequence's tokens equal the
    single-process graph decoder's on the same GPU."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world, n_ticks = 2, 14
    out = str(tmp_path / "hist.npy")
    mp.spawn(_pipeline_worker, args=(world, 29561, n_ticks, out), nprocs=world, join=True)
    hist = np.load(out)
    from exllamav2_amd.cache import ExLlamaV2Cache
    from exllamav2_amd.model import ExLlamaV2, GreedyGraphDecoder
    from exllamav2_amd.synth import synth_checkpoint
    cfg = _cfg()
    model = ExLlamaV2(cfg, device="cuda:0").load(synth_checkpoint(cfg, "cuda:0", seed=5))
    for s, tok0 in enumerate([3, 11, 40, 77]):
        cache = ExLlamaV2Cache(model, batch_size=1, max_seq_len=256)
        dec = GreedyGraphDecoder(model, cache, batch_size=1).capture()
        dec.reset(torch.tensor([tok0]), 0)
        dec.run(3)
        torch.cuda.synchronize()
        want = dec.tokens(0, 3).cpu().numpy()[0]
        dec.free()
        assert np.array_equal(hist[s, 1:4], want), (s, hist[s, :5], want)
    model.unload()


def _tp_worker(rank, world, port, out_dir):
    _init(rank, world, port)
    from exllamav2_amd.cache import ExLlamaV2Cache
    from exllamav2_amd.synth import synth_checkpoint
    from exllamav2_amd.tensor_p import ExLlamaV2TP, TPGreedyDecoder
    cfg = _cfg()
    model = ExLlamaV2TP(cfg, rank, world, device="cuda:0").load(synth_checkpoint(cfg, "cuda:0", seed=9))
    cache = ExLlamaV2Cache(model, batch_size=1, max_seq_len=256)
    logits = model.forward(torch.tensor([PROMPT]), cache, last_id_only=False)
    dec = TPGreedyDecoder(model, cache, batch_size=1)
    dec.reset(torch.tensor([int(torch.argmax(logits[0, -1]))]), len(PROMPT))
    dec.run(N_DECODE)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"logits{rank}.npy"), logits.float().cpu().numpy())
    np.save(os.path.join(out_dir, f"tokens{rank}.npy"), dec.tokens(len(PROMPT), N_DECODE).cpu().numpy())
    model.unload()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_tensor_parallel_two_ranks_real_library(tmp_path):
    """column shards + gathers, two ranks on cuda:0: ranks agree bit for bit, logits within the model tolerance of the oracle,
    greedy tokens equal where the oracle is confident."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = 2
    mp.spawn(_tp_worker, args=(world, 29563, str(tmp_path)), nprocs=world, join=True)
    l0, l1 = (np.load(tmp_path / f"logits{r}.npy") for r in range(world))
    t0, t1 = (np.load(tmp_path / f"tokens{r}.npy") for r in range(world))
    assert np.array_equal(l0, l1) and np.array_equal(t0, t1)
    from exllamav2_amd.synth import synth_checkpoint
    from oracle.model import OracleModel
    cfg = _cfg()
    oracle = OracleModel(cfg, synth_checkpoint(cfg, "cuda:0", seed=9))      # (the device generator's stream, like the ranks')
    oracle.reset(1)
    want = oracle.forward(np.array([PROMPT]))
    assert np.abs(l0.astype(np.float64) - want).max() < LOGIT_TOL
    tok = int(np.argmax(l0[0, -1]))
    checked = 0
    for i in range(N_DECODE):
        w = oracle.forward(np.array([[tok]]))[0, -1]
        top = np.sort(w)[-2:]
        if top[1] - top[0] > 4 * LOGIT_TOL:
            assert int(t0[0, i]) == int(np.argmax(w))
            checked += 1
        tok = int(t0[0, i])
    assert checked >= 1, "vacuous token check"


def _run_bench(extra, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    if env: e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, env=e,
                          cwd=ROOT, timeout=600)


def test_bench_gpus_n_without_the_devices_fails_loudly():
    """`python bench.py --gpus N` with fewer than N visible GPUs (this container: 0; a one-GPU box: 1) must exit non-zero with a
    message and print NO JSON line -- an n_gpus = 1 line would be read as the N-GPU figure."""
    n = max(2, torch.cuda.device_count() + 1)
    r = _run_bench(["--gpus", str(n), "--steps", "2", "--warmup", "1", "--headline-only", "--no-parity-check"])
    assert r.returncode != 0
    assert f"--gpus {n}" in r.stderr and "visible" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_refuses_a_world_size_that_is_not_gpus():
    r = _run_bench(["--gpus", "4", "--steps", "2"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
