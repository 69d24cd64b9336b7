"""tts-king hot path (FastSpeech2 train step + HiFi-GAN generator inference) on MI355X / gfx950.

Python host code on PyTorch-ROCm (device memory, streams, torch.distributed) calling hand-written HIP kernels
through the C ABI in include/ttsk.h.  No CPU fallback: see tts_king_amd/lib.py.
"""
__version__ = "0.1.0"
