#!/bin/bash
# Builds a copy of libvrt.so from a git revision's sources (default HEAD) as raytracing_amd/ab_<name>.so, for
# A/B runs against the working tree's build (tools/ab_libs.sh).  usage: tools/build_variant.sh name [rev] [EXTRA flags]
set -e
name=$1; rev=${2:-HEAD}; extra=${3:-}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/vrt_variant.XXXXXX)
git -C "$root" archive "$rev" raytracing_amd/csrc include | tar -x -C "$tmp"
make -s -j4 -C "$tmp/raytracing_amd/csrc" ../libvrt.so EXTRA="$extra"
cp "$tmp/raytracing_amd/libvrt.so" "$root/raytracing_amd/ab_$name.so"
rm -rf "$tmp"
echo "raytracing_amd/ab_$name.so <- $rev $extra"
