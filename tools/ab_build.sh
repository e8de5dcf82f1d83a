#!/bin/bash
# Builds an alternative libbdpt_amd.so (+ the host library linked against it) into build/ab/<name>/ with extra
# hipcc flags, for timing two builds in one GPU session:  BDPT_LIB_DIR=build/ab/<name> python tools/prof_run.py ...
#   tools/ab_build.sh <name> [-DFLAG=...] ...
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
OUT=build/ab/$NAME
mkdir -p "$OUT"
PKG=opencl-path-tracer_amd
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function "$@" \
      -shared -x hip $PKG/csrc/bdpt_api.cpp $PKG/csrc/bdpt_group.cpp $PKG/csrc/bdpt_kernels.hip -o "$OUT/libbdpt_amd.so" -ldl
g++ -O2 -std=c++17 -fPIC -Wall -shared $PKG/host/camera.cpp $PKG/host/scenes.cpp $PKG/host/renderer.cpp $PKG/host/scene_io.cpp \
    $PKG/host/image_io.cpp $PKG/host/capi.cpp -o "$OUT/libbdpt_host.so" -L"$OUT" -lbdpt_amd -Wl,-rpath,'$ORIGIN'
echo "built $OUT"
