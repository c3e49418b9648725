"""peppan_amd - MI355X-native replacement for PEPPAN's similarity-search hot path
(modules/uberBlast.py + modules/clust.py of zheminzhou/PEPPAN).  HIP kernels live in csrc/,
reached through the C ABI of include/peppan_hip.h via ctypes (_native.py)."""
__version__ = '0.1.0'
