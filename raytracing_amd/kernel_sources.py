"""Identity of the trace kernels' sources: the git blob hash (sha1 of "blob <len>\\0" + content, what `git hash-object`
prints) of every file the kernels are compiled from.  A counter record (profiles/traffic.json) carries the hashes of
the sources it was measured on; bench.py prices a run with the record only when they are the hashes of the tree it
runs from."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ("raytracing_amd/csrc/vrt_trace.hip", "raytracing_amd/csrc/vrt_kernels.h", "raytracing_amd/csrc/vrt_device.h")


def blob_hash(path):
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def blob_hashes(root=ROOT):
    return {rel: blob_hash(os.path.join(root, rel)) for rel in KERNEL_SOURCES}
