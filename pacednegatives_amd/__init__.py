"""MI355X-native implementation of the PacedNegatives lceT5 training hot path.

Public surface mirrors the reference package for that path:
``pacednegatives_amd.lceT5`` (LCEModel, LCEDataModule, LCEWeights, ChangeDifficulty),
``pacednegatives_amd.dataloader`` (LCEDataset and the pair/level loaders) and
``pacednegatives_amd.mining`` (dense negative-pool selection).  All device arithmetic runs in
hand-written HIP kernels behind the C ABI of ``include/paced_hip.h``.
"""
__version__ = "0.1.0"
