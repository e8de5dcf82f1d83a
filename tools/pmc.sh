#!/bin/bash
# Collects hardware counters for the render kernel, one rocprofv3 --pmc pass per
# counter group (a pass can hold 8 SQ, 4 TCC and 2 GRBM counters), and prints one
# line per counter for the timed launches.  Usage (on the GPU box):
#   tools/pmc.sh <outdir> [prof_run.py arguments]
set -u
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
 "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32"
 "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_LEVEL_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE GRBM_COUNT"
 "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_IFETCH"
 "FETCH_SIZE"
 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -s KILL 180 rocprofv3 --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 tools/prof_run.py "$@" > "$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$OUT/pass$i.log"; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = collections.OrderedDict()
for f in sorted(glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "bdpt" not in r["Kernel_Name"] or "query" in r["Kernel_Name"]:
            continue
        per[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in per.items():
        rows[(k, c)] = v
with open(out + "/summary.txt", "w") as fo:
    for (k, c), v in rows.items():
        line = "%-50s %-28s last=%.6g n=%d" % (k[-50:], c, v[-1], len(v))
        print(line); fo.write(line + "\n")
PY
