"""gcn_vae_amd: MI355X-native R-GCN-VAE link-prediction hot path (hand-written HIP kernels behind a
C ABI) with host modules that keep the names and signatures of karenyang/GCN-VAE's kgvae/ package.

The compute path needs the built ``libgcnvae_hip.so`` and a ROCm device; there is no CPU fallback.
"""
__version__ = '0.1.0'
