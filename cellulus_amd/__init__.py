"""cellulus_amd — MI355X-native hot path of funkelab/cellulus.

Mirrors the reference's Python API (``cellulus.train``, ``cellulus.infer``,
``cellulus.models.get_model``, ``cellulus.criterions.get_loss``, the config
classes) on top of hand-written HIP kernels for gfx950 (``libclx.so``, C ABI in
``include/clx.h``).  There is no CPU compute path.
"""

__version__ = "0.1.0"
