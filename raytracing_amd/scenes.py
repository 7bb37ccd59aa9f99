"""Benchmark / parity configurations: window geometry and camera poses.

Window geometry follows the reference: the loaded area is the cube of N voxels
per axis centred on the middle of the player's chunk
(src/graphics/shaders/rendering/raytrace.glsl:19-24), and world voxel p lives
at texel rem_euclid(p, N) (src/terrain/terrain_gen/terrain_gen.comp:23).
"""
import math

import numpy as np

from .player import Player

CHUNK_SIZE = 32  # src/graphics/mod.rs:147


def chunk_of(position):
    """`GameState::current_player_chunk` (src/game_logic/game_state.rs:20-22)."""
    return tuple(int(math.floor(np.float32(c) / np.float32(CHUNK_SIZE))) for c in position)


def window_min(position, n):
    """Lowest world voxel of the loaded cube for a camera at `position`."""
    return tuple(c * CHUNK_SIZE + CHUNK_SIZE // 2 - n // 2 for c in chunk_of(position))


def grid_n_for_render_distance(render_distance):
    """src/graphics/mod.rs:359: image extent = 32 * (2*render_distance + 3)."""
    return CHUNK_SIZE * (2 * render_distance + 3)


# Pose A: the spawn pose (identity).  Degenerate: one pixel column has rd.x == 0.
def pose_spawn():
    return Player.spawn()


# Pose B: generic pose, no direction component is exactly zero.
POSE_B_POSITION = (3.37, 14.71, -5.13)
POSE_B_YAW = 0.6
POSE_B_PITCH = -0.35


def pose_b(y=None):
    pos = list(POSE_B_POSITION)
    if y is not None:
        pos[1] = y
    return Player.at(pos, POSE_B_YAW, POSE_B_PITCH)


def pose_headline():
    """Camera of the headline metric (1080p over 1024^3): pose B raised to y = 40.3
    so that the horizon is in frame."""
    return pose_b(40.3)


def flythrough_poses(count=64, radius=300.0, height=60.3, pitch=-0.3):
    """Camera poses on a circle: yaw_k = 2*pi*k/count + 0.013."""
    out = []
    for k in range(count):
        a = 2.0 * math.pi * k / count
        out.append(Player.at((radius * math.cos(a), height, radius * math.sin(a)), a + 0.013, pitch))
    return out


def cam16(player):
    """Column-major 16 x f32 of the player's transform (the shader's mat4)."""
    return player.push_constants().column_major()
