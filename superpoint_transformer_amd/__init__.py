"""MI355X-native hot path of Superpoint Transformer (segment-CSR scatter,
sparse superpoint-graph attention, radius-kNN + point geometric features).

Python here is host plumbing over ``libspt_hip.so`` (hand-written gfx950 HIP
kernels behind the C ABI of ``include/spt_hip.h``).
"""
__version__ = "0.1.0"
