"""meshdqn_amd - MI355X-native hot path of MeshDQN (IPCS flow solve + graph Q-network).

Python host code on PyTorch-ROCm calling hand-written gfx950 HIP kernels
through the C ABI in include/meshdqn_hip.h.  There is no CPU fallback: the
compute classes raise `MeshDQNHipError` when the HIP library or a GPU is missing.
"""
import os as _os

# Streams that are meant to run BESIDE each other (env step / IPCS step of the previous env step / optimiser step / mesh
# mirrors) must not share a hardware queue: the HIP runtime deals its streams round-robin over GPU_MAX_HW_QUEUES (default
# 4) queues and kernels of one queue run in submission order.  Read by the runtime when it initialises (first HIP call).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from ._lib import MeshDQNHipError  # noqa: E402,F401

__version__ = "0.1.0"
