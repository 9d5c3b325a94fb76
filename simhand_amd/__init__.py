"""simhand_amd -- MI355X-native (gfx950) implementation of SiMHand's contrastive
pre-training hot path: hand-written HIP kernels behind a C ABI
(``include/simhand_hip.h`` / ``libsimhand_hip.so``) plus the Python host mirror
of the reference's step classes and CLI (``simhand_amd.host``).

There is no CPU execution path in this package; the CPU oracle lives under
``oracle/`` and is test infrastructure only.
"""
__version__ = "0.1.0"
