"""CPU oracle for the EXL2/GPTQ quantized forward path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (`exllamav2_amd/`) may import this
package; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use it, and
only as the checker.

Every function is a restatement (numpy, plain loops for tiny cases) of the reference's semantics and
cites the reference file:line it follows.  Parity status:

* **pinned by execution of the reference's own code**: the packed K-stream bit order of every width
  (2, 3, 4, 5, 6, 8 bits), the 4-bit scale decode (`dq_scale`) and the GPTQ `(q - zero)` decode --
  `oracle/ref_build/` compiles the reference's `exllamav2_ext/cuda/quant/qdq_*.cuh` for the host from
  where they lie under /root/reference (a shim supplies the CUDA fp16 vocabulary; outputs in
  `oracle/_ref/`, git-ignored), `tests/golden/make_golden_qdq.py` records what the reference's
  load-time shuffle + kernel-side dequant return for seeded inputs, and `tests/test_oracle_ref.py`
  checks `oracle/exl2.py` against that fixture everywhere and against the live library here;
  the quantized-KV-cache codec (`cuda/cache_q.cuh` run with 256 logical threads per block on a host
  fiber scheduler, `oracle/ref_build/simt_host.*`; fixture `tests/golden/reference_cache_q.npz`,
  generator `tests/golden/make_golden_cacheq.py`): codes, scales, dequantized values bit for bit; the
  paged / contiguous addressing kernels of `cuda/cache.cu` around it
  (`tests/golden/reference_cache_addressing.npz`);
  `reconstruct()` itself (the text of `shuffle_kernel` + `reconstruct_kernel` extracted from
  `cuda/q_matrix.cu` at build time into the git-ignored `oracle/_ref/`, run block by block with threads
  as fibers; fixture `tests/golden/reference_reconstruct.npz`): `exl2_reconstruct` bit for bit on every
  width / mix / act-order case;
  the decode GEMV kernel `gemm_half_q_half_kernel` (header template, included as it lies) as the
  measured yardstick for the q_gemm tolerance (`tests/golden/reference_q_gemm.npz`);
  the RoPE rotation (`rope_cuda_arr_neox` / `_gptj` of `cuda/rope.cu`, `tests/golden/reference_rope.npz`) and
  `rms_norm_kernel` (`cuda/rms_norm.cu`; the float64-sum oracle is within one fp16 ulp of it); the MoE
  routing kernels of `cuda/q_mlp_softmax.cuh` (`tests/golden/reference_moe_routing.npz`);
  the CPU sampler (`cpp/sampling.cpp`, plain C++: compiled as it lies and driven in `sample_basic`'s order,
  `tests/golden/reference_sampling.npz`): `oracle/sampling.py` bit for bit -- tokens, probabilities, candidate counts;
  likewise the pure torch functions of the reference that run on CPU (group map, RMSNorm, attention,
  RoPE tables, MLP activation: `tests/golden/make_golden.py`);
* **unpinned by execution, pinned by relation**: the multiply itself (the reference has no CPU q_gemm
  and ships no golden vectors; its semantics are `matmul(x, reconstruct())`:
  `gemm(I) == reconstruct()`, one-hot rows at full size, `gemm(x) ~ x @ reconstruct()`, with the
  reference kernel's own error as the yardstick) and the fused module kernels (compositions).
"""
