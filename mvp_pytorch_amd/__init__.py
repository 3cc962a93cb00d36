"""mvp_pytorch_amd — MI355X (gfx950) implementation of MVPTR's cross-modal BERT encoder path.

Host side mirrors the reference's `oscar.modeling.*` class surface (see mvp_pytorch_amd.modeling);
the encoder itself is hand-written HIP in csrc/ behind the C ABI of include/mvptr.h.
"""
__version__ = "0.1.0"

