"""vae_amd -- MI355X-native (gfx950) implementation of the scVI-style VAE training hot path
of DiseaseNeuroGenomics/vae: hand-written HIP kernels behind a C ABI (include/vae_amd.h),
with the reference's Python module API on top."""
__version__ = "0.1.0"
