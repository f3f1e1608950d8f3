"""MI355X-native offline-MPC hot path (batched condensed QP + structured NN forward).

Host side mirrors the reference's Python interface
(pratyushkumar211/industrial_nnmpc_2021, lib/linearMPC.py,
lib/LinearMPCLayers.py, lib/controller_evaluation.py); the arithmetic runs in
hand-written HIP kernels behind the C ABI declared in ``include/nnmpc.h``
(``libnnmpc_hip.so``, built by ``__graft_entry__.build()``).  There is no CPU
fallback: using a solver without the library or without a GPU raises.
"""
__version__ = "0.1.0"
