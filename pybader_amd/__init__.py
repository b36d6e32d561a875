"""pybader_amd -- MI355X-native drop-in for the hot path of pybader v0.3.12.

Only what the neargrid/ongrid assignment + edge-refinement path needs lives here:

    csrc/            hand-written HIP kernels + the C ABI (include/bader_hip.h) -> libbader_hip.so
    _lib.py          ctypes binding (no PyTorch); raises if the library or a GPU is missing
    thread_handlers  bader_calc / refine / assign_to_atoms / dtype_calc   (pybader/thread_handlers.py)
    methods          ongrid / neargrid kernel-level plugins               (pybader/methods.py)
    refinement       neargrid / edge_find / edge_check                    (pybader/refinement.py)
    utils            vacuum_assign / charge_sum / atom_assign / ...       (pybader/utils.py, njit half)
    interface        a minimal Bader counterpart with the reference's attribute and method names
    slab             z-slab (axis-0) scheduler for several GPUs, one process per GPU
    synth            bit-reproducible synthetic densities for tests and bench.py
"""
__version__ = '0.1.0'
