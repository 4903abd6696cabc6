"""CPU oracle for the EXL2/GPTQ quantized forward path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (`exllamav2_amd/`) may import this
package; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use it, and
only as the checker.

Every function is a restatement (numpy, plain loops for tiny cases) of the reference's semantics and
cites the reference file:line it follows.  Parity status: **unpinned by execution** -- the reference
has no CPU implementation of this path and ships no golden vectors, so the
oracle is pinned by *relation* (pack -> unpack round trips that restate both `pack_tensor.cu` and the
plain `qdq_*.cuh` decoders, identity-matrix GEMM == reconstruct, WHT involution) and by the pure
torch functions of the reference that do run on CPU (`tests/golden/make_golden.py` imports those and
commits their outputs as fixtures).
"""
