"""The C++ host mirror (include/graphics_core.hpp) driven like the reference's main.rs by
examples/flythrough.cpp: spawn, 5^3 chunk upload, per-frame update_player + redraw, chunk-window
moves with partial regeneration.  The last frame must equal the oracle's frame of the world the
same chunk policy produces on the CPU."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "flythrough")


def test_flythrough_last_frame_matches_oracle(tmp_path):
    assert os.path.exists(EXE), "build with make -C examples (done by __graft_entry__.build())"
    w, h, frames = 320, 240, 150
    out = tmp_path / "frame.bgra"
    res = subprocess.run([EXE, str(frames), str(w), str(h), str(out)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    lines = res.stdout.strip().splitlines()
    cam = np.array([float(v) for v in [l for l in lines if l.startswith("transform")][0].split()[1:]], np.float32)
    moves = [tuple(int(v) for v in l.split()[1:]) for l in lines if l.startswith("chunk")]
    assert len(moves) >= 2, "the flight should cross chunk boundaries"
    # replay GameState::upload_chunks_around_player on the CPU (src/game_logic/game_state.rs:50-83)
    n, gen = 160, 2
    block = np.zeros((n, n, n), np.uint16)
    cur = (0, 0, 0)
    for x in range(-gen, gen + 1):
        for y in range(-gen, gen + 1):
            for z in range(-gen, gen + 1):
                O.terrain_chunk(block, x, y, z)
    for new in moves:
        for x in range(-gen, gen + 1):
            for y in range(-gen, gen + 1):
                for z in range(-gen, gen + 1):
                    cp = (new[0] + x, new[1] + y, new[2] + z)
                    if max(abs(a - b) for a, b in zip(cp, cur)) <= gen:
                        continue
                    O.terrain_chunk(block, *cp)
        cur = new
    want = O.trace(O.World(block), cam, w, h, want=("bgra8",))["bgra8"]
    got = np.fromfile(out, np.uint8).reshape(h, w, 4)
    assert np.array_equal(got, want)
    assert (want[..., :3] != 0).any()
