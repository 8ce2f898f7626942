"""pos_evolution_amd: MI355X-native attestation aggregation + LMD-GHOST fork choice.

The product is ``libposevo.so`` (HIP kernels + C ABI, ``include/posevo.h``); this package is
the thin Python host layer above it:

    _abi        ctypes declarations of the C ABI
    engine      numpy-level wrapper, one method per pe_* entry point
    forkchoice  the reference's interface (get_head / on_attestation / process_attestation ...)
    sharded     validator-range sharding over N GPUs (torch.distributed / RCCL exchange)

Import never falls back to a CPU implementation: without the built library it raises.
"""
from . import _abi
from .engine import RESIDENT, ROWS_RESIDENT, AttRow, DeviceArena, DeviceRows, Engine, EngineError, pack_attestations

__all__ = ["Engine", "EngineError", "AttRow", "DeviceArena", "DeviceRows", "pack_attestations", "RESIDENT", "ROWS_RESIDENT",
           "_abi"]
