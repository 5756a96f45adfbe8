"""sclens_amd -- MI355X (gfx950) implementation of the scLENS `sclens()` hot path.

`sclens_amd.api` mirrors the reference's operator interface over the C ABI of `libsclens_hip.so` (include/sclens_hip.h);
`sclens_amd.synth` generates the seeded synthetic count matrices of the benchmark; `sclens_amd.shard` holds the
multi-GPU plumbing (independent decompositions over ranks), `sclens_amd.atlas` the row-sharded variant for cells > genes.
There is no CPU fallback: every call needs the built library and an AMD GPU.
"""
__version__ = "0.1.0"
