"""elba_amd — MI355X-native overlap-detection engine for ELBA (k-mer counting -> A -> B = A*A^T, SharedSeeds semiring).

The product is libelba_amd.so (hand-written HIP for gfx950 behind the C ABI of include/elba_amd.h); this package is a thin
ctypes binding used by the tests, the benchmark and the multi-GPU driver.  There is no CPU fallback: importing works anywhere,
but creating an engine without the built library or without a GPU raises.
"""
from .capi import (ElbaError, Engine, Seed, SEED_DTYPE, OVERLAP_DTYPE, lib_path, load_library, synth_reads, SynthCfg)  # noqa: F401

__all__ = ["ElbaError", "Engine", "Seed", "SEED_DTYPE", "OVERLAP_DTYPE", "lib_path", "load_library", "synth_reads", "SynthCfg"]
