// stub: the CUDA vocabulary comes from oracle/ref_build/cuda_shim.h (force-included)
