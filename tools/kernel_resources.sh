#!/bin/bash
# Prints register, spill, scratch and LDS usage of every kernel in libbdpt_amd.so's gfx950 code object.
set -e
cd "$(dirname "$0")/.."
TMP=$(mktemp -d)
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Iinclude \
      --cuda-device-only -S -o "$TMP/k.s" opencl-path-tracer_amd/csrc/bdpt_kernels.hip
python3 - "$TMP/k.s" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", txt, re.S):
    pass
for blk in re.split(r"\n  - \.agpr_count", txt)[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    print("%-70s vgpr %s sgpr %s vspill %s sspill %s scratch %s lds %s" % (
        name[:70], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
        g("private_segment_fixed_size"), g("group_segment_fixed_size")))
PY
rm -rf "$TMP"
