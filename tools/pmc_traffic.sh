#!/bin/bash
# HBM traffic of the render kernel for one workload: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes.
#   tools/pmc_traffic.sh <outdir> [prof_run.py arguments]
set -u
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 tools/prof_run.py "$@" > "$OUT/fetch.log" 2>&1 || { echo "FETCH_SIZE pass failed"; tail -3 "$OUT/fetch.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 tools/prof_run.py "$@" > "$OUT/write.log" 2>&1 || { echo "WRITE_SIZE pass failed"; tail -3 "$OUT/write.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "bdpt" in r["Kernel_Name"] and "query" not in r["Kernel_Name"]:
            print(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"], r["Counter_Value"])
PY
