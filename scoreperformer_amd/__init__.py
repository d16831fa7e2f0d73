"""MI355X-native implementation of the ScorePerformer transformer hot path.

`scoreperformer_amd.modules` / `scoreperformer_amd.models` mirror `scoreperformer.modules` / `scoreperformer.models`;
every forward/backward runs hand-written HIP kernels for gfx950 through the C-ABI library `libspn.so`.
"""
__version__ = "0.1.0"
