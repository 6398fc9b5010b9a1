"""Seeded synthetic LaserScan streams for tests, golden fixtures and bench.py.

The reference ships no rosbags or fixtures, so every input
in this repo is generated here from ``numpy.random.default_rng(seed)``.

World and sensor follow SURVEY.md section 8(d):

* world: axis-aligned room (10 m x 8 m, scaled by ``room_scale``) with six
  circular pillars of radius 0.3 m, ray-cast analytically;
* sensor: the Gazebo ray sensor of the reference robot
  (course_agv_description/urdf/course_agv.gazebo:36-58): beam angles
  ``linspace(-3.14159, 3.14159, N)``, range clip [0.10, 30] m, Gaussian range
  noise sigma = 0.01 m, ranges delivered as float32 (the LaserScan wire type),
  ``inf`` where nothing is hit;
* trajectory: unicycle at v = 0.3 m/s, 10 Hz, along a seeded smooth closed curve
  inside the room (turn rate within about +-0.5 rad/s), so a replay of any length
  stays clear of walls and pillars.

Pure NumPy; no GPU, no reference code.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

import numpy as np

ANGLE_MIN = -3.14159
ANGLE_MAX = 3.14159
RANGE_MIN = 0.10
RANGE_MAX = 30.0
RANGE_SIGMA = 0.01

_PILLARS = (
    (-3.0, -2.0), (-1.0, 2.5), (1.5, -2.5), (3.5, 1.0), (0.5, 0.5), (-3.5, 1.5),
)
_PILLAR_R = 0.3


@dataclass
class LaserScan:
    """Duck-type of ``sensor_msgs/LaserScan`` with the fields the hot path reads
    (W7 icp.py:190-193, W12 slam_ekf.py:116-121): ``ranges``, ``angle_min``,
    ``angle_max``.  ``ranges`` is a tuple of Python floats holding float32 values,
    exactly what rospy hands to a callback."""

    ranges: Sequence[float]
    angle_min: float = ANGLE_MIN
    angle_max: float = ANGLE_MAX
    angle_increment: float = 0.0
    range_min: float = RANGE_MIN
    range_max: float = RANGE_MAX
    header: object = None


@dataclass
class World:
    half_x: float = 5.0
    half_y: float = 4.0
    pillars: Tuple[Tuple[float, float], ...] = _PILLARS
    pillar_r: float = _PILLAR_R

    @staticmethod
    def room(scale: float = 1.0) -> "World":
        return World(5.0 * scale, 4.0 * scale,
                     tuple((x * scale, y * scale) for x, y in _PILLARS), _PILLAR_R * scale)


def raycast(world: World, px: np.ndarray, py: np.ndarray, ang: np.ndarray) -> np.ndarray:
    """Exact distance along rays (px,py)+t*(cos ang, sin ang) to the first wall or
    pillar.  Shapes broadcast; returns float64, ``inf`` if nothing is hit."""
    px, py, ang = np.broadcast_arrays(px, py, ang)
    c, s = np.cos(ang), np.sin(ang)
    best = np.full(px.shape, np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        for wall, o, d, other_o, other_d, lim in (
            (world.half_x, px, c, py, s, world.half_y), (-world.half_x, px, c, py, s, world.half_y),
            (world.half_y, py, s, px, c, world.half_x), (-world.half_y, py, s, px, c, world.half_x),
        ):
            t = (wall - o) / d
            hit = other_o + t * other_d
            ok = (t > 1e-9) & np.isfinite(t) & (np.abs(hit) <= lim + 1e-9)
            best = np.where(ok & (t < best), t, best)
        for cx, cy in world.pillars:
            ox, oy = px - cx, py - cy
            b = ox * c + oy * s
            disc = b * b - (ox * ox + oy * oy - world.pillar_r ** 2)
            t = -b - np.sqrt(np.where(disc >= 0, disc, np.nan))
            ok = (disc >= 0) & (t > 1e-9)
            best = np.where(ok & (t < best), t, best)
    return best


def trajectory(world: World, n: int, seed: int, v: float = 0.3, dt: float = 0.1) -> np.ndarray:
    """Unicycle poses [n,3] (x, y, theta): constant speed ``v`` along a seeded closed
    curve (an ellipse around the room centre whose radius is modulated by three
    random low-frequency harmonics), heading = path tangent.  The turn rate stays
    within about +-0.5 rad/s and the curve keeps >= 0.6 m from every pillar and wall
    of ``World.room``; theta is continuous (not wrapped), as the reference's dead
    reckoning is (W7 icp.py:158)."""
    rng = np.random.default_rng(seed)
    scale = world.half_x / 5.0
    amp = rng.uniform(-1.0, 1.0, size=3) * np.array([0.10, 0.06, 0.03])
    ph = rng.uniform(0.0, 2.0 * np.pi, size=3)
    phi0 = rng.uniform(0.0, 2.0 * np.pi)
    sense = 1.0 if rng.uniform() < 0.5 else -1.0

    def curve(phi):
        m = 1.0 + amp[0] * np.sin(2 * phi + ph[0]) + amp[1] * np.sin(3 * phi + ph[1]) + amp[2] * np.sin(5 * phi + ph[2])
        return 2.3 * scale * m * np.cos(phi), 1.6 * scale * m * np.sin(phi)

    # arc-length parametrisation on a fine grid, then constant-speed sampling
    laps = int(np.ceil(n * v * dt / (10.0 * scale))) + 2
    fine = np.linspace(0.0, 2.0 * np.pi * laps, 20000 * laps)
    fx, fy = curve(phi0 + sense * fine)
    seg = np.hypot(np.diff(fx), np.diff(fy))
    arc = np.concatenate([[0.0], np.cumsum(seg)])
    s = np.arange(n) * v * dt
    u = np.interp(s, arc, fine)
    x, y = curve(phi0 + sense * u)
    h = 1e-5
    x2, y2 = curve(phi0 + sense * (u + h))
    th = np.unwrap(np.arctan2(y2 - y, x2 - x))
    return np.stack([x, y, th], axis=1)


def scans_from_poses(world: World, poses: np.ndarray, n_beams: int, seed: int,
                     noise: float = RANGE_SIGMA) -> np.ndarray:
    """float32 ranges [n_scan, n_beams] seen from ``poses`` (sensor at the robot
    origin, course_agv.gazebo:33)."""
    rng = np.random.default_rng(seed + 7919)
    beam = np.linspace(ANGLE_MIN, ANGLE_MAX, n_beams)
    r = raycast(world, poses[:, 0:1], poses[:, 1:2], poses[:, 2:3] + beam[None, :])
    r = r + rng.normal(0.0, noise, size=r.shape)
    r = np.where(r > RANGE_MAX, np.inf, np.maximum(r, RANGE_MIN))
    return r.astype(np.float32)


@dataclass
class Replay:
    """A synthetic scan stream: ``ranges`` float32 [n_scan, n_beams] plus the
    ground-truth poses they were cast from."""

    ranges: np.ndarray
    poses_true: np.ndarray
    angle_min: float = ANGLE_MIN
    angle_max: float = ANGLE_MAX
    seed: int = 0
    room_scale: float = 1.0
    messages: List[LaserScan] = field(default_factory=list, repr=False)

    def message(self, k: int) -> LaserScan:
        return LaserScan(ranges=tuple(float(v) for v in self.ranges[k]),
                         angle_min=self.angle_min, angle_max=self.angle_max,
                         angle_increment=(self.angle_max - self.angle_min) / (self.ranges.shape[1] - 1))


def make_replay(n_scans: int, n_beams: int = 360, seed: int = 1, room_scale: float = 1.0,
                noise: float = RANGE_SIGMA, stride: int = 1) -> Replay:
    """SURVEY.md 8(d) cfg2/cfg4/cfg5 style replay (seed 1: cfg2; 10-17: cfg4; 3 with
    ``room_scale=2, n_beams=1080``: cfg5).  ``stride`` keeps every stride-th message of the
    10 Hz stream, as the reference's callbacks do before processing a scan (every 5th in
    W12m/slam_ekf.py:65-68, every 6th in W7/icp.py:51-54): ``n_scans`` is the number of
    PROCESSED scans, spaced ``stride * 0.1`` s apart."""
    world = World.room(room_scale)
    poses = trajectory(world, n_scans * stride, seed)[::stride]
    ranges = scans_from_poses(world, poses, n_beams, seed, noise)
    return Replay(ranges=ranges, poses_true=poses, seed=seed, room_scale=room_scale)


def scan_pair(n_beams: int = 360, seed: int = 0, delta=(0.05, 0.02, np.deg2rad(1.0)),
              shape: str = "room", noise: float = RANGE_SIGMA) -> Replay:
    """Two scans of a static world from poses ``p`` and ``p (+) delta`` (cfg1)."""
    rng = np.random.default_rng(seed)
    if shape == "room":
        world = World.room(1.0)
    elif shape == "corridor":
        world = World(12.0, 1.2, (), 0.0)
    elif shape == "circle":
        world = None  # robot inside a circular wall of radius 6 m, handled below
    else:
        raise ValueError(shape)
    p0 = np.array([rng.uniform(-1.0, 1.0), rng.uniform(-0.6, 0.6), rng.uniform(-np.pi, np.pi)])
    if shape == "circle":
        p0[:2] *= 0.5
    c, s = np.cos(p0[2]), np.sin(p0[2])
    p1 = np.array([p0[0] + c * delta[0] - s * delta[1], p0[1] + s * delta[0] + c * delta[1], p0[2] + delta[2]])
    poses = np.stack([p0, p1])
    if shape == "circle":
        beam = np.linspace(ANGLE_MIN, ANGLE_MAX, n_beams)
        ang = poses[:, 2:3] + beam[None, :]
        ox, oy = poses[:, 0:1], poses[:, 1:2]
        b = ox * np.cos(ang) + oy * np.sin(ang)
        r = -b + np.sqrt(b * b - (ox * ox + oy * oy - 36.0))
        r = r + np.random.default_rng(seed + 7919).normal(0.0, noise, size=r.shape)
        ranges = np.maximum(r, RANGE_MIN).astype(np.float32)
    else:
        ranges = scans_from_poses(world, poses, n_beams, seed, noise)
    return Replay(ranges=ranges, poses_true=poses, seed=seed)


def particle_priors(n_particles: int, seed: int = 2, sigma_xy: float = 0.05,
                    sigma_th: float = np.deg2rad(2.0)) -> np.ndarray:
    """cfg3 prior perturbations [P,3]: (dx, dy) ~ N(0, 0.05^2), dtheta ~ N(0, (2 deg)^2)."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n_particles, 3))
    out[:, 0:2] = rng.normal(0.0, sigma_xy, size=(n_particles, 2))
    out[:, 2] = rng.normal(0.0, sigma_th, size=n_particles)
    return out
