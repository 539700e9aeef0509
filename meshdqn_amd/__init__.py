"""meshdqn_amd - MI355X-native hot path of MeshDQN (IPCS flow solve + graph Q-network).

Python host code on PyTorch-ROCm calling hand-written gfx950 HIP kernels
through the C ABI in include/meshdqn_hip.h.  There is no CPU fallback: the
compute classes raise `MeshDQNHipError` when the HIP library or a GPU is missing.
"""
from ._lib import MeshDQNHipError  # noqa: F401

__version__ = "0.1.0"
