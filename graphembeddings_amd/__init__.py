"""graphembeddings_amd -- MI355X-native drop-in for the hot path of greysun/GraphEmbeddings' holE.py.

Only what that path needs: `csrc/` (hand-written HIP kernels + the C ABI of include/ge_hip.h),
`hole` (host mirror of the reference's operator interface), `data` (file formats), `sharded`
(row-sharded multi-GPU step) and `train` (the reference's driver loop and CLI flags).
"""
__version__ = "0.1.0"
