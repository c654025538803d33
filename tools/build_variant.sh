#!/bin/bash
# tools/build_variant.sh <name> [device flags...]: the library with other device-code flags, as ab/<name>.so (for tools/abperf.sh).
# Host objects must be up to date (make -C raytracer_rs_amd/csrc first).
set -e
NAME=$1; shift
cd "$(dirname "$0")/../raytracer_rs_amd/csrc"
mkdir -p ../../ab
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-builtin-sin -fno-builtin-cos -Wall -Wextra -Wno-unused-parameter"
/opt/rocm/bin/hipcc $FLAGS --offload-arch=gfx950 "$@" -c rt_device.hip -o /tmp/rt_device_$NAME.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../ab/$NAME.so rt_capi.o rt_multi.o rt_flatten.o rt_loader.o rt_toml.o rt_canvas.o rt_jpeg.o /tmp/rt_device_$NAME.o -ldl -lpthread
