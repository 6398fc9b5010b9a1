"""Parameter lookup with the reference's names.

The reference reads its settings with ``rospy.get_param(name, default)``
(W12m/icp.py:14-25,40; W12m/slam_ekf.py:22-32).  Here the same names are served from a
process-local table (``set_param``), falling back to a live ROS parameter server when
``rospy`` is importable, else to the caller's default, so the drop-in classes need no ROS.
"""
from __future__ import annotations

_PARAMS = {}
_MISSING = object()


def set_param(name, value):
    _PARAMS[name] = value


def clear_params():
    _PARAMS.clear()


def get_param(name, default=_MISSING):
    if name in _PARAMS:
        return _PARAMS[name]
    try:
        import rospy  # noqa: WPS433 (optional)
        if default is _MISSING:
            return rospy.get_param(name)
        return rospy.get_param(name, default)
    except Exception:
        if default is _MISSING:
            raise KeyError(name)
        return default
