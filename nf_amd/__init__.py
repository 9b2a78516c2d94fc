"""nf_amd: MI355X-native (gfx950) neural importance sampler, drop-in for the hot path of NGoetz/NF
(`nisrep`): PWLin/PWQuad coupling flows, variance-loss training, Monte-Carlo integration and the
RAMBO-on-diet phase-space map.  Kernels live in nf_amd/csrc (HIP) behind the C ABI of include/nis_hip.h."""
__version__ = "0.1.0"
