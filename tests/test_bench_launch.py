"""`python bench.py --gpus N` started plainly (no launcher, no WORLD_SIZE): the parent starts one rank per GPU
through torch.distributed.run as a child process, relays rank 0's JSON line and exits with the child's code."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, env=env, timeout=timeout)
    return p


def json_lines(text):
    return [json.loads(l) for l in text.splitlines() if l.lstrip().startswith("{")]


def test_plain_start_with_two_ranks_launches_them_itself():
    p = run_bench("--gpus", "2", "--steps", "2", "--launch-check")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = json_lines(p.stdout)
    assert len(lines) == 1, p.stdout
    assert lines[0]["n_gpus"] == 2 and lines[0]["ranks_seen"] == 2 and lines[0]["steps"] == 2


def test_rank_count_that_disagrees_with_gpus_is_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=3" in p.stderr and not p.stdout.strip()


@pytest.mark.gpu
@pytest.mark.parametrize("pixels,steps", [("bgr", "6"), ("bgra", "2")])  # six steps: a whole buffer of four frames and a partial one
def test_two_ranks_on_one_gpu_over_gloo_print_one_line(pixels, steps):
    """The whole multi-rank bench (replicated worlds, one pose per rank, framebuffer gather to rank 0), started
    plainly, with both ranks sharing the one GPU of the box and the gather going through host memory.  Rank 0 checks
    the last frame that arrived from every rank against that rank's pose traced by itself."""
    p = run_bench("--gpus", "2", "--backend", "gloo", "--steps", steps, "--warmup", "1", "--no-cpu-baseline", "--probe", "16",
                  "--gather-pixels", pixels, "--gather-every", "4")
    assert p.returncode == 0, p.stderr[-3000:]
    lines = json_lines(p.stdout)
    assert len(lines) == 1, p.stdout
    assert lines[0]["n_gpus"] == 2 and lines[0]["steps"] == int(steps) and lines[0]["value"] > 0
    assert lines[0]["scaling"] == "weak" and lines[0]["roofline"]["timed_launches"] >= 16
    assert lines[0]["host_enqueue_ms_per_step"] > 0
    assert lines[0]["gather"] == {"bytes_per_pixel_on_the_link": 3 if pixels == "bgr" else 4, "frames_per_gather": 4,
                                  "last_frame_of_every_rank_equals_its_trace": True}


@pytest.mark.gpu
def test_single_rank_over_rccl_gathers_packed_pixels():
    """The N > 1 code path with RCCL itself, as far as one GPU goes: a process group of one rank, the pack kernel, the
    gather on the context's stream, the unpack kernels behind it."""
    p = run_bench("--gpus", "1", "--force-gather", "--steps", "9", "--warmup", "2", "--no-cpu-baseline", "--no-literal", "--probe", "16")
    assert p.returncode == 0, p.stderr[-3000:]
    lines = json_lines(p.stdout)
    assert len(lines) == 1 and lines[0]["gather"]["last_frame_of_every_rank_equals_its_trace"] is True
    assert lines[0]["parity"]["bgra8_sha256_equals_oracle_golden"] is True


@pytest.mark.gpu
@pytest.mark.parametrize("pixels", ["bgr", "bgra"])
def test_strip_sharded_frame_single_rank_over_rccl(pixels):
    """`--shard strips` through RCCL as far as one GPU goes: the rank's strips packed (three bytes per pixel unless bgra),
    gathered on the context's stream, put back in row order (with the alpha byte again) by the reorder kernel; bench.py
    itself compares the reassembled frame with the frame traced whole, and the line with the oracle's frozen frame."""
    p = run_bench("--gpus", "1", "--force-gather", "--shard", "strips", "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
                  "--no-literal", "--probe", "16", "--gather-pixels", pixels)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = json_lines(p.stdout)
    assert len(lines) == 1 and lines[0]["scaling"] == "strong"
    assert lines[0]["gather"]["bytes_per_pixel_on_the_link"] == (3 if pixels == "bgr" else 4)
    assert lines[0]["parity"]["bgra8_sha256_equals_oracle_golden"] is True


@pytest.mark.gpu
def test_strip_sharded_frame_two_ranks_over_gloo():
    """The same partition with two ranks on the one GPU (gloo, through host memory): parts of 544 and 536 rows, padded,
    three bytes per pixel; rank 0's reassembled frame is the frame traced whole (asserted by bench.py) and the oracle's."""
    p = run_bench("--gpus", "2", "--backend", "gloo", "--shard", "strips", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                  "--probe", "16")
    assert p.returncode == 0, p.stderr[-3000:]
    lines = json_lines(p.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["scaling"] == "strong"
    assert lines[0]["gather"] == {"bytes_per_pixel_on_the_link": 3, "bytes_per_frame_on_the_links": 544 * 1920 * 3}
    assert lines[0]["parity"]["bgra8_sha256_equals_oracle_golden"] is True


def test_counter_record_the_bench_prices_is_complete():
    """bench.py takes the kernel's bytes, instruction counts and counters per launch from profiles/traffic.json (committed
    counter passes of its own command): both launch modes must be there, and the figures must hang together."""
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for key in ("skip", "skip_valu_instructions", "skip_salu_instructions", "skip_plain_order", "skip_plain_order_valu_instructions"):
        assert isinstance(t[key], (int, float)) and t[key] > 0, key
    for mode in ("skip_counters", "skip_plain_order_counters"):
        c = t[mode]
        assert abs(t[mode[:-len("_counters")]] - (c["fetch_bytes"] + c["write_bytes"])) < 1
        assert c["write_bytes"] >= 1920 * 1080 * 4             # at least the frame is written
        assert 0.5 < c["lanes_active_per_vector_instruction"] <= 1.0
        assert 0.0 < c["l1_hit_rate"] <= 1.0 and 0.0 < c["l2_hit_rate"] <= 1.0
        assert 0.0 < c["issue_wait_fraction_of_wave_cycles"] < 1.0 and 0.0 < c["memory_wait_fraction_of_wave_cycles"] < 1.0
        assert c["waves"] == 32400   # 60 x 540 wavefront tiles of 32 x 2 pixels
    assert 0.9 < t["skip_plain_order_valu_instructions"] / t["skip_valu_instructions"] < 1.0


def test_counter_record_was_measured_on_the_kernel_sources_of_this_tree():
    """The record carries the git blob hashes of the files the kernel was compiled from when it was measured; bench.py
    prices a run with it only when they match the tree's (else it prints valu_count_stale and no fraction).  A change
    to the kernels without a fresh counter pass fails here."""
    from raytracing_amd.kernel_sources import blob_hashes
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert t["kernel_sources"] == blob_hashes(), "run tools/profile_r03.sh on the GPU box and copy its traffic.json to profiles/"
