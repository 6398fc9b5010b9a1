"""rospy adapter: run the package's nodes under ROS 1 with the topics, frames
and message types of the reference's scripts.

    python -m <package>.ros_node icp            # W7/icp.py:202-205          node "icp_node"
    python -m <package>.ros_node slam           # W12m/slam_ekf.py:277-281   node "slam_node"
    python -m <package>.ros_node slam_online    # W12o/slam_ekf.py           (+4 evidence, /tf centre)
    python -m <package>.ros_node localization   # W9/localization.py:246-250 node "localization_node"

The node classes themselves never import ROS: they hand plain dicts to ``publish`` hooks.  This
module is the only place that does; it turns those dicts into ``nav_msgs/Odometry`` /
``nav_msgs/OccupancyGrid`` messages and subscribes the callbacks.  ROS is not installed in the
build image, so the adapter is exercised in the tests with stand-in ``rospy`` / message modules
that record what is published (tests/test_gpu_ros_adapter.py).
"""
from __future__ import annotations

import importlib
import sys


def _ros():
    """The ROS modules, imported on demand so that the package stays importable without them."""
    names = ("rospy", "tf", "nav_msgs.msg", "sensor_msgs.msg", "tf2_msgs.msg")
    try:
        return {n: importlib.import_module(n) for n in names}
    except ImportError as e:                                      # pragma: no cover - depends on the host
        raise ImportError("the ROS adapter needs ROS 1 python packages (%s)" % e)


class OdometryPublisher:
    """``publish(dict)`` -> nav_msgs/Odometry on ``topic`` (fields as W7/icp.py:166-180)."""

    def __init__(self, ros, topic):
        self._ros = ros
        self._pub = ros["rospy"].Publisher(topic, ros["nav_msgs.msg"].Odometry, queue_size=3)

    def publish(self, odom):
        m = self._ros["nav_msgs.msg"].Odometry()
        m.header.stamp = self._ros["rospy"].Time.now()
        m.header.frame_id = odom["frame_id"]
        p, q = odom["position"], odom["orientation"]
        m.pose.pose.position.x, m.pose.pose.position.y, m.pose.pose.position.z = p
        (m.pose.pose.orientation.x, m.pose.pose.orientation.y, m.pose.pose.orientation.z,
         m.pose.pose.orientation.w) = q
        self._pub.publish(m)


class TransformBroadcaster:
    """``sendTransform(translation, rotation, time, child, parent)`` with time=None -> now."""

    def __init__(self, ros):
        self._ros = ros
        self._br = ros["tf"].TransformBroadcaster()

    def sendTransform(self, translation, rotation, time, child, parent):
        self._br.sendTransform(translation, rotation, time if time is not None else self._ros["rospy"].Time.now(), child, parent)


class MapPublisher:
    """``publish(dict)`` -> nav_msgs/OccupancyGrid on ``/slam_map`` (W12m/slam_ekf.py:252-275); the
    int8 data arrive from the device already in the wire layout."""

    def __init__(self, ros, topic="/slam_map"):
        self._ros = ros
        self._pub = ros["rospy"].Publisher(topic, ros["nav_msgs.msg"].OccupancyGrid, queue_size=1)

    def publish(self, grid):
        m = self._ros["nav_msgs.msg"].OccupancyGrid()
        m.header.stamp = self._ros["rospy"].Time.now()
        m.header.frame_id = grid["frame_id"]
        m.info.resolution = grid["resolution"]
        m.info.width, m.info.height = grid["width"], grid["height"]
        m.info.origin.position.x, m.info.origin.position.y, m.info.origin.position.z = grid["origin"]
        m.info.origin.orientation.x = m.info.origin.orientation.y = m.info.origin.orientation.z = 0
        m.info.origin.orientation.w = 1.0
        m.data = grid["data"].tolist()
        self._pub.publish(m)


class ScanPublisher:
    """W9's ``/target_laser``: the virtual scan is a copy of the incoming LaserScan message with
    other ranges, so it can be published as it is."""

    def __init__(self, ros, topic="/target_laser"):
        self._pub = ros["rospy"].Publisher(topic, ros["sensor_msgs.msg"].LaserScan, queue_size=3)

    def publish(self, msg):
        self._pub.publish(msg)


def make_icp(ros=None):
    from .icp import ICP
    ros = ros or _ros()
    ros["rospy"].init_node("icp_node")
    node = ICP()
    node.odom_pub = OdometryPublisher(ros, "icp_odom")
    node.odom_broadcaster = TransformBroadcaster(ros)
    node.laser_sub = ros["rospy"].Subscriber("/course_agv/laser/scan", ros["sensor_msgs.msg"].LaserScan, node.laserCallback)
    return node


def make_slam(ros=None, online=False, landmarks=True):
    from .slam_ekf import SLAM_EKF
    ros = ros or _ros()
    ros["rospy"].init_node("slam_node")
    node = SLAM_EKF(online=online, landmarks=landmarks)
    node.map_pub = MapPublisher(ros)
    node.laser_sub = ros["rospy"].Subscriber("/course_agv/laser/scan", ros["sensor_msgs.msg"].LaserScan, node.laserCallback)
    if online:
        node.tf_sub = ros["rospy"].Subscriber("/tf", ros["tf2_msgs.msg"].TFMessage, node.tf_callback)
    return node


def make_localization(ros=None):
    from .localization import Localization
    ros = ros or _ros()
    ros["rospy"].init_node("localization_node")
    node = Localization()
    node.location_pub = OdometryPublisher(ros, "ekf_w8")
    node.location_pub1 = OdometryPublisher(ros, "ekf_w9")
    node.odom_pub = OdometryPublisher(ros, "icp_odom")
    node.odom_broadcaster = TransformBroadcaster(ros)
    node.laser_pub = ScanPublisher(ros)
    node.laser_sub = ros["rospy"].Subscriber("/course_agv/laser/scan", ros["sensor_msgs.msg"].LaserScan, node.laserCallback)
    node.map_sub = ros["rospy"].Subscriber("/map", ros["nav_msgs.msg"].OccupancyGrid, node.updateMap)
    return node


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    which = argv[0] if argv else "slam"
    ros = _ros()
    if which == "icp":
        make_icp(ros)
    elif which == "slam":
        make_slam(ros)
    elif which == "slam_online":
        make_slam(ros, online=True)
    elif which == "localization":
        make_localization(ros)
    else:
        raise SystemExit("usage: ros_node {icp|slam|slam_online|localization}")
    ros["rospy"].spin()


if __name__ == "__main__":
    main()
