"""neuralplda_amd — MI355X-native (gfx950) Neural-PLDA scoring/training hot path.

The arithmetic lives in libnplda_hip.so (C ABI: include/nplda_hip.h; sources: csrc/*.hip);
this package is the host-side mirror of the reference's Python interface for that path.
"""
__version__ = "0.1.0"
