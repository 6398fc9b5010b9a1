"""Importable alias of the package directory (whose name, fixed by the project layout,
contains hyphens): ``import course_agv_slam_amd as slam; slam.ICP()``."""
import importlib as _il
import sys as _sys

_pkg = _il.import_module("a-2d-lidar-based-slam-system-for-wheeled-mobile-robots_amd")
_sys.modules[__name__] = _pkg
