"""bmnas — MI355X-native engine under the BM-NAS ``models.search.darts`` module surface.

``bmnas.lib``     ctypes binding of libbmnas_hip.so (the C ABI in include/bmnas_hip.h)
``bmnas.cell``    forward/backward procedures of the fusion cell built from those kernels
``bmnas.functions`` torch.autograd.Function wrappers used by the nn.Module mirror
``bmnas.dist``    RCCL data parallelism (replaces nn.DataParallel)
"""
__all__ = ['lib', 'cell', 'functions', 'dist', 'build']
