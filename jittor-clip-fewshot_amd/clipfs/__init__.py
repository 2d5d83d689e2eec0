"""clipfs: MI355X (gfx950) engine for the CLIP few-shot hot path.

``_lib``   ctypes binding of libclipfs_hip.so (C ABI: include/clipfs.h)
``ops``    tensor-level wrappers (one HIP kernel chain each)
``engine`` tower runtime, autograd plumbing, differentiable head ops
``synth``  synthetic weights / inputs (no network: pretrained checkpoints are absent)
``safe_pkl`` inert reader for Jittor-saved pickles
"""
