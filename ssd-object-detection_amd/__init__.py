"""MI355X-native SSD hot path (gfx950 HIP kernels behind a C ABI) with the reference's Python
surface on top: utils.bbox, models.ssd_model, data_loaders.ssd, tools.train.

The HIP library is mandatory: nothing here falls back to a CPU implementation."""
__version__ = "0.1.0"
