"""nerfslam -- host-side mirror of the reference's hot-path interfaces (Python on PyTorch-ROCm).

Everything numerical runs in the hand-written HIP kernels of libnerfslam_hip.so; this package
only allocates tensors, keeps the factor-graph bookkeeping and forwards to the C ABI.
"""
