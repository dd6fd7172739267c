"""Drop-ins for the two third-party CUDA packages the reference imports but does not vendor:

    knn_cuda.KNN                                      (KNN_CUDA 0.2 wheel, registration/Dockerfile:8)
    pointnet2.utils.pointnet2_utils.grouping_operation (Pointnet2_PyTorch @5ff4382, registration/README.md:20)

`install()` registers them in sys.modules under those names so that the UNMODIFIED reference files
registration/src/soft_projection.py and samplenet.py import and run on MI355X with the kernels of
libsamplenet_hip.so underneath (level 1).
"""
import sys
import types


def install():
    from . import knn_cuda
    from .pointnet2.utils import pointnet2_utils

    sys.modules.setdefault("knn_cuda", knn_cuda)
    p2 = types.ModuleType("pointnet2")
    p2u = types.ModuleType("pointnet2.utils")
    p2.utils = p2u
    p2u.pointnet2_utils = pointnet2_utils
    sys.modules.setdefault("pointnet2", p2)
    sys.modules.setdefault("pointnet2.utils", p2u)
    sys.modules.setdefault("pointnet2.utils.pointnet2_utils", pointnet2_utils)
