"""MI355X-native scan matching + occupancy mapping for the course_agv SLAM stack.

Drop-in replacements, on hand-written gfx950 HIP kernels behind a C ABI
(include/slam_hip.h, csrc/), for the per-scan hot path of
zjwzcx/A-2D-LiDAR-based-SLAM-System-for-Wheeled-Mobile-Robots:

    ICP        process / findNearest / getTransform / laserToNumpy / laserCallback / publishResult
    Mapping    update -> pmap
    bresenham  (start, end).path
    SLAM_EKF   laserCallback glue (scan matching + map building); landmarks=True: the whole W12
               node with Extraction and the landmark EKF on the host
    Localization  updateMap / laserEstimation / calc_map_observation (scan-to-map, W9)

plus the batched forms used by bench.py (``replay``) and the multi-GPU sharding helper
(``dist``).  Importing the package never computes anything; every operator raises if
libslamhip.so or the GPU is missing (there is no CPU implementation in the product).
"""
from . import _abi, dist, param, synthetic
from ._abi import Context, LibraryMissing, SlamError, default_context
from .bresenham import bresenham, rasterize
from .ekf_lm import EKF
from .extraction import Extraction, LandMarkSet
from .icp import ICP, scan_to_pc
from .localization import Localization
from .mapping import Mapping
from .replay import DeviceGrid, DeviceReplay, icp_batch_host, particles_host, prior_matrices, replay_host
from .slam_ekf import SLAM_EKF
from .synthetic import LaserScan

__all__ = ["ICP", "Mapping", "Localization", "EKF", "Extraction", "LandMarkSet", "bresenham", "rasterize", "SLAM_EKF", "LaserScan", "Context", "default_context",
           "DeviceGrid", "DeviceReplay", "replay_host", "icp_batch_host", "particles_host", "prior_matrices", "scan_to_pc", "SlamError",
           "LibraryMissing", "param", "synthetic"]
