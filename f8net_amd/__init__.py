"""f8net_amd — MI355X-native fixed-point-8 integer inference path (the `int_op_only` forward of
snap-research/F8Net) behind a C-ABI shared library of hand-written gfx950 HIP kernels.

Layout: csrc/ (HIP kernels + C ABI -> libf8net.so), _lib (ctypes binding), net (graph builder /
planner front end), ops (op-level seam), int_model (IntModel-shaped modules), dist (batch sharding),
topology / synth (net tables, deterministic synthetic parameters).
"""
__version__ = '0.1.0'
