"""Camera pose -> the 4x4 f32 transform the trace path consumes.

Host-side mirror of the reference's `Player` (src/game_logic/player/mod.rs:5-64)
and of `PushConstants` (src/graphics/push_constants.rs:4-9).  The trace path only
needs the matrix; this module exists so tests and benchmarks can build the same
matrices the game would hand to `GraphicsCore::redraw`.
"""
import math
from dataclasses import dataclass, field

import numpy as np

f32 = np.float32


def _rotation_from_scaled_axis(v):
    """nalgebra `Matrix4::from_scaled_axis`: rotation by |v| radians about v/|v|
    (Rodrigues form), identity for the zero vector.  f32 arithmetic throughout."""
    v = np.asarray(v, dtype=f32)
    angle = f32(math.sqrt(float(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])))
    m = np.eye(4, dtype=f32)
    if angle == 0:
        return m
    ux, uy, uz = (v / angle).astype(f32)
    s, c = f32(math.sin(float(angle))), f32(math.cos(float(angle)))
    omc = f32(1) - c
    sqx, sqy, sqz = ux * ux, uy * uy, uz * uz
    m[:3, :3] = np.array([
        [sqx + (f32(1) - sqx) * c, ux * uy * omc - uz * s, ux * uz * omc + uy * s],
        [ux * uy * omc + uz * s, sqy + (f32(1) - sqy) * c, uy * uz * omc - ux * s],
        [ux * uz * omc - uy * s, uy * uz * omc + ux * s, sqz + (f32(1) - sqz) * c]], dtype=f32)
    return m


@dataclass
class PushConstants:
    """`#[repr(C)] struct PushConstants { transform: Matrix4<f32>, image_index: u32 }`
    (src/graphics/push_constants.rs:4-9).  `transform` is 4x4, indexed [row, col];
    `column_major()` gives the 16 floats in the order the shader's mat4 holds them."""
    transform: np.ndarray
    image_index: int = 0

    def column_major(self):
        return np.ascontiguousarray(np.asarray(self.transform, dtype=f32).T).reshape(16)


@dataclass
class Player:
    """src/game_logic/player/mod.rs:5-64."""
    matrix: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=f32))
    look_direction: tuple = (0.0, 0.0)  # (yaw, pitch)

    BASE_SPEED = 0.02

    @classmethod
    def spawn(cls):  # mod.rs:15-20
        return cls()

    @classmethod
    def at(cls, position, yaw, pitch):
        p = cls()
        p.matrix[:3, 3] = np.asarray(position, dtype=f32)
        p.look_direction = (float(yaw), float(pitch))
        p.apply_rotation()
        return p

    def get_transform(self):  # mod.rs:12-14
        return self.matrix.copy()

    def get_position(self):  # mod.rs:22-24
        return self.matrix[:3, 3].copy()

    def apply_rotation(self):  # mod.rs:25-36: matrix = T * Rx(pitch) * Ry(yaw)
        t = self.get_position()
        yaw, pitch = self.look_direction
        rot = _rotation_from_scaled_axis([pitch, 0, 0]) @ _rotation_from_scaled_axis([0, yaw, 0])
        m = np.eye(4, dtype=f32)
        m[:3, 3] = t
        self.matrix = (m @ rot).astype(f32)

    def update_pos(self, keys, speed_modifier=1.0):  # mod.rs:37-63
        yaw = self.look_direction[0]
        forward = np.array([-math.sin(yaw), 0.0, math.cos(yaw)], dtype=f32)
        right = np.array([math.cos(yaw), 0.0, math.sin(yaw)], dtype=f32)
        up = np.array([0.0, 1.0, 0.0], dtype=f32)
        mov = np.zeros(3, dtype=f32)
        if "W" in keys: mov += forward
        if "A" in keys: mov -= right
        if "S" in keys: mov -= forward
        if "D" in keys: mov += right
        if "Space" in keys: mov += up
        if "LShift" in keys: mov -= up
        cap = f32(self.BASE_SPEED * speed_modifier)
        norm = f32(np.sqrt((mov * mov).sum(dtype=f32)))
        if norm > cap:
            mov = mov * (cap / norm)
        self.matrix[:3, 3] += mov

    def push_constants(self):
        return PushConstants(self.get_transform(), 0)
