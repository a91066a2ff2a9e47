"""emdr2_amd -- MI355X-native hot path of EMDR2 (retriever -> MIPS over the evidence index -> reader).

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all compute on the
path runs in hand-written HIP kernels behind the C ABI declared in include/*.h (libemdr2_hip.so).
There is no CPU fallback: importing the native layer without the built library raises.
"""
__version__ = "0.1.0"
