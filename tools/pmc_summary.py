"""Condenses one tools/pmc.sh run (summary.txt of per-kernel counters) into the entry bench.py quotes as
roofline.executed: the hardware's view of one launch of the render kernel.

    python tools/pmc_summary.py <summary.txt> <key> <kernel_ms> <source path as committed> [>> merged into profiles/pmc_summary.json]

executed_tflops = FP32 vector flops the lanes actually performed / kernel time:
  (2 * FMA + MUL + ADD wave-instructions) * 64 lanes * lane utilisation,
  lane utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)   (both count 4-cycle units over all VALU work).
valu_issue_busy = VALU wave-instructions * 2 cycles / (SIMDs * kernel clocks), kernel clocks = GRBM_GUI_ACTIVE / 8 XCDs.
simd_cycles_per_valu_inst = SIMDs * kernel clocks / VALU wave-instructions (tools/valu_issue.hip: 1.36 for back-to-back
fma at six wavefronts per SIMD); wavefronts_per_simd_mean = SQ_WAVE_CYCLES (4-cycle units) * 4 / (SIMDs * kernel clocks).
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path, key, kernel_ms, source = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
c = {}
for line in open(path):
    m = re.search(r"(\S+)\s+last=(\S+)", line)
    if m:
        c[m.group(1)] = float(m.group(2))
lane_util = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
fp_insts = c["SQ_INSTS_VALU_FMA_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c["SQ_INSTS_VALU_ADD_F32"]
flops = (2 * c["SQ_INSTS_VALU_FMA_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c["SQ_INSTS_VALU_ADD_F32"]) * 64 * lane_util
clocks = c["GRBM_GUI_ACTIVE"] / 8.0
entry = {
    "source": source,
    "kernel_ms_under_profiler": kernel_ms,
    "valu_wave_insts": c["SQ_INSTS_VALU"],
    "fp32_arith_fraction_of_valu": round(fp_insts / c["SQ_INSTS_VALU"], 3),
    "lane_utilisation": round(lane_util, 3),
    "executed_tflops": round(flops / (kernel_ms * 1e-3) / 1e12, 2),
    "executed_frac_of_157.3": round(flops / (kernel_ms * 1e-3) / 1e12 / 157.3, 3),
    "valu_issue_busy": round(c["SQ_INSTS_VALU"] * 2.0 / (1024 * clocks), 3),
    "simd_cycles_per_valu_inst": round(1024 * clocks / c["SQ_INSTS_VALU"], 2),
    "wavefronts_per_simd_mean": round(c["SQ_WAVE_CYCLES"] * 4.0 / (1024 * clocks), 2),
    "salu_per_valu": round(c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"], 3) if "SQ_INSTS_SALU" in c else None,
    "branch_per_valu": round(c["SQ_INSTS_BRANCH"] / c["SQ_INSTS_VALU"], 3) if "SQ_INSTS_BRANCH" in c else None,
    "lds_array_busy": round(c["SQ_LDS_IDX_ACTIVE"] / (256 * clocks), 3) if "SQ_LDS_IDX_ACTIVE" in c else None,
    "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT"),
    "wave_cycles_waiting_fraction": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
    "fetch_kb": c.get("FETCH_SIZE"), "write_kb": c.get("WRITE_SIZE"),
}
out = os.path.join(ROOT, "profiles", "pmc_summary.json")
try:
    allv = json.load(open(out))
except (OSError, ValueError):
    allv = {"note": "per-launch hardware counters of the render kernel (tools/pmc.sh passes condensed by tools/pmc_summary.py); "
                    "bench.py quotes the entry of its workload and kernel as roofline.executed"}
allv[key] = entry
json.dump(allv, open(out, "w"), indent=1)
print(json.dumps(entry, indent=1))
