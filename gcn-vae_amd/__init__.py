"""gcn_vae_amd: MI355X-native R-GCN-VAE link-prediction hot path (hand-written HIP kernels behind a
C ABI) with host modules that keep the names and signatures of karenyang/GCN-VAE's kgvae/ package.

The compute path needs the built ``libgcnvae_hip.so`` and a ROCm device; there is no CPU fallback.
"""
__version__ = '0.1.0'

# ``ops`` is one namespace over three files (ops.py, indices.py, made.py, which import each other): loading it here, in its own order,
# makes ``from gcn_vae_amd import made`` / ``indices`` safe as a first import too.
from . import ops  # noqa: E402,F401
