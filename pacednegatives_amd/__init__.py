"""MI355X-native implementation of the PacedNegatives lceT5 training hot path.

Public surface mirrors the reference package for that path:
``pacednegatives_amd.lceT5`` (LCEModel, LCEDataModule, LCEWeights, ChangeDifficulty),
``pacednegatives_amd.dataloader`` (LCEDataset and the pair/level loaders) and
``pacednegatives_amd.mining`` (dense negative-pool selection).  All device arithmetic runs in
hand-written HIP kernels behind the C ABI of ``include/paced_hip.h``.
"""
import os as _os

# Kernel arguments in device memory instead of host memory: a training step is ~2500 launches, and with the arguments in
# host memory the first workgroup of a launch on every CU waits for them across the bus (3.5 k cycles against 1 k,
# tools/kernarg_latency.hip; later workgroups hit the scalar cache either way).  Worth 0.7-0.9 ms of the benchmark's 116 ms
# step.  The HIP runtime reads the variable when it initialises, i.e. at the first GPU call of the process, so it is set
# here, at import; a value already in the environment is left alone.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

__version__ = "0.1.0"
