"""smoothxg_amd -- MI355X-native blocked partial-order-alignment engine (smoothxg hot path).

Only what the path needs: csrc/ (HIP kernels + the C ABI of include/sxg_poa.h), the ctypes
host mirror (poa.py), the synthetic workload generator (synth.py) and block sharding
(shard.py).  The CPU oracle lives in oracle/ and is never imported from here.
"""
import os as _os

# one launch per geometry on its own stream: give the HIP runtime enough hardware queues for them to
# run side by side (read when the runtime initialises; see sxg_poa_create)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from .poa import PoaEngine, PoaError, Params, params_from_cli, load_library, xxh64  # noqa: F401
