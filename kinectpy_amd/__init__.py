"""kinectpy_amd -- MI355X-native point-cloud preprocessing behind KinectPy's
preprocessing.{extractor,filtering,registration} and floor_removal module APIs.

Layout: csrc/ (HIP kernels + the C ABI of include/kinectpx.h), _lib/ops (ctypes binding, tensor
operators), geometry/o3d (the Open3D surface the path touches), preprocessing/, floor_removal,
utils/ (host-side mirrors of the reference modules), pipeline/parallel (multi-sensor frame loop).
"""
__version__ = "0.1.0"
