"""The rospy adapter wired to stand-in ROS modules that record what is
subscribed and published: topics, message types and message contents of the reference's nodes."""
import types

import numpy as np
import pytest

from conftest import load_golden, pkg

pytestmark = pytest.mark.gpu
AMIN, AMAX = -3.14159, 3.14159


class Rec:
    """Attribute bag standing in for a ROS message."""
    def __init__(self):
        object.__setattr__(self, "_d", {})

    def __getattr__(self, k):
        d = object.__getattribute__(self, "_d")
        if k not in d:
            d[k] = Rec()
        return d[k]

    def __setattr__(self, k, v):
        object.__getattribute__(self, "_d")[k] = v


def fake_ros():
    log = {"pub": [], "sub": [], "tf": [], "nodes": []}

    class Publisher:
        def __init__(self, topic, typ, queue_size=1):
            self.topic, self.typ = topic, typ
        def publish(self, m):
            log["pub"].append((self.topic, m))

    class Subscriber:
        def __init__(self, topic, typ, cb):
            log["sub"].append((topic, typ.__name__, cb))

    class Time:
        @staticmethod
        def now():
            return 123.0

    class Br:
        def sendTransform(self, *a):
            log["tf"].append(a)

    def msgtype(name):
        return type(name, (Rec,), {})

    ros = {"rospy": types.SimpleNamespace(Publisher=Publisher, Subscriber=Subscriber, Time=Time, spin=lambda: None,
                                          init_node=lambda n: log["nodes"].append(n)),
           "tf": types.SimpleNamespace(TransformBroadcaster=Br),
           "nav_msgs.msg": types.SimpleNamespace(Odometry=msgtype("Odometry"), OccupancyGrid=msgtype("OccupancyGrid")),
           "sensor_msgs.msg": types.SimpleNamespace(LaserScan=msgtype("LaserScan")),
           "tf2_msgs.msg": types.SimpleNamespace(TFMessage=msgtype("TFMessage"))}
    return ros, log


def scan(slam, r):
    return slam.LaserScan(ranges=tuple(float(v) for v in r), angle_min=AMIN, angle_max=AMAX,
                          angle_increment=(AMAX - AMIN) / (len(r) - 1))


def test_icp_node(syn):
    slam = pkg()
    ros, log = fake_ros()
    node = pkg("ros_node").make_icp(ros)
    assert log["nodes"] == ["icp_node"] and log["sub"][0][:2] == ("/course_agv/laser/scan", "LaserScan")
    rep = syn.make_replay(8, 120, seed=4, stride=1)
    cb = log["sub"][0][2]
    for k in range(8):
        cb(rep.message(k))
    (topic, m), = log["pub"]                                   # first scan = target, then every 6th message
    assert topic == "icp_odom" and m.header.frame_id == "world_base" and m.header.stamp == 123.0
    assert m.pose.pose.position.x == node.sensor_sta[0] and m.pose.pose.position.z == 0.001
    assert m.pose.pose.orientation.w == pytest.approx(np.cos(node.sensor_sta[2] / 2))
    assert log["tf"][0][3:] == ("icp_odom", "world_base") and log["tf"][0][2] == 123.0


def test_slam_node_publishes_the_map():
    slam = pkg()
    g7 = load_golden("g7_w12_node.npz")
    ros, log = fake_ros()
    node = pkg("ros_node").make_slam(ros)
    cb = log["sub"][0][2]
    for r in g7["node_ranges"][:20]:
        cb(scan(slam, r))
    maps = [m for t, m in log["pub"] if t == "/slam_map"]
    assert len(maps) == 3                                       # messages 10, 15 and 20 (the 5th is the first scan)
    m = maps[-1]
    assert (m.info.width, m.info.height, m.info.resolution, m.header.frame_id) == (200, 200, 0.1, "map")
    assert m.info.origin.position.x == -10.0 and m.info.origin.orientation.w == 1.0
    data = np.array(m.data, dtype=np.int8).reshape(200, 200)    # data[y*width + x] = pmap[x][y]
    assert np.array_equal(data.T, node.mapping.pmap.astype(np.int8))


def test_online_and_localization_wiring():
    ros, log = fake_ros()
    pkg("ros_node").make_slam(ros, online=True)
    assert [s[:2] for s in log["sub"]] == [("/course_agv/laser/scan", "LaserScan"), ("/tf", "TFMessage")]
    ros, log = fake_ros()
    node = pkg("ros_node").make_localization(ros)
    assert [s[0] for s in log["sub"]] == ["/course_agv/laser/scan", "/map"]
    g5 = load_golden("g5_map_observation.npz")
    node.obstacle = g5["obs_wall"]
    slam = pkg()
    cb = log["sub"][0][2]
    for k in range(6):
        cb(scan(slam, g5["obs_ranges"][0]))
    topics = [t for t, _ in log["pub"]]
    assert topics == ["ekf_w8", "ekf_w9", "icp_odom", "/target_laser"]
    assert [a[3] for a in log["tf"]] == ["ekf_w8", "ekf_w9", "icp_odom"]
