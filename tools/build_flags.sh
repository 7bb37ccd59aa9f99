#!/bin/bash
# Builds the working tree's library with extra compiler flags as raytracing_amd/ab_<name>.so (for A/B runs).
# usage: tools/build_flags.sh name "<EXTRA flags>"
set -e
name=$1; extra=${2:-}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/vrt_flags.XXXXXX)
mkdir -p "$tmp/raytracing_amd" "$tmp/include"
cp -r "$root/raytracing_amd/csrc" "$tmp/raytracing_amd/"; cp "$root/include/vrt.h" "$tmp/include/"
rm -f "$tmp"/raytracing_amd/csrc/*.o
make -s -j3 -C "$tmp/raytracing_amd/csrc" ../libvrt.so EXTRA="$extra"
cp "$tmp/raytracing_amd/libvrt.so" "$root/raytracing_amd/ab_$name.so"
rm -rf "$tmp"
echo "raytracing_amd/ab_$name.so <- working tree, $extra"
