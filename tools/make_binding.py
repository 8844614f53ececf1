#!/usr/bin/env python3
"""tools/make_binding.py <out.csv> <pmc_summary.csv>... -- what binds one kernel, from the per-kernel counter summaries of
tools/pmc.sh (separate rocprofv3 --pmc passes): the raw per-launch counters of the kernel matching KERNEL, and below them
the derived occupancy of every unit in fractions of the kernel's cycles.  SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles per wave or per SIMD (x 4 = cycles); SQ_LDS_* count cycles per CU (MI355X_MICROARCH.md, "PMC units")."""
import csv
import sys

KERNEL = "bwd_prim_kernel"
out, files = sys.argv[1], sys.argv[2:]
c, disp = {}, 0
for f in files:
    for line in open(f).read().splitlines()[1:]:
        kernel, dispatches, counter, _, per = line.rsplit(",", 4)   # (template arguments put commas into the name)
        if KERNEL in kernel and "precise" not in kernel:
            c[counter] = float(per)
            disp, name = int(dispatches), kernel
cyc = c["GRBM_GUI_ACTIVE"] / 8.0                 # summed over the 8 XCDs -> kernel cycles
simds, cus = 1024, 256
rows = [("kernel", name), ("dispatches_averaged", disp), ("kernel_cycles (GRBM_GUI_ACTIVE / 8 XCDs)", cyc),
        ("kernel_ms_at_2.4GHz", cyc / 2.4e6)]
wave_cyc = c["SQ_WAVE_CYCLES"] * 4
d = [
    ("waves_per_SIMD (SQ_WAVE_CYCLES x 4 / (1024 SIMDs x cycles))", wave_cyc / (simds * cyc)),
    ("VALU_busy (SQ_ACTIVE_INST_VALU x 4 / (1024 x cycles))", c["SQ_ACTIVE_INST_VALU"] * 4 / (simds * cyc)),
    ("LDS_pipe_busy (SQ_LDS_IDX_ACTIVE / (256 CUs x cycles))", c["SQ_LDS_IDX_ACTIVE"] / (cus * cyc)),
    ("LDS_bank_conflict_share (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)", c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]),
    ("LDS_issue_busy (SQ_ACTIVE_INST_LDS x 4 / (1024 x cycles))", c["SQ_ACTIVE_INST_LDS"] * 4 / (simds * cyc)),
    ("scalar_issue_busy (SQ_ACTIVE_INST_SCA x 4 / (1024 x cycles))", c.get("SQ_ACTIVE_INST_SCA", 0) * 4 / (simds * cyc)),
    ("LDS_data_fifo_full_share_of_CU_cycles", c.get("SQ_LDS_DATA_FIFO_FULL", 0) / (cus * cyc)),
    ("wave_time: issuing (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES)", c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]),
    ("wave_time: parked at s_waitcnt / barrier (SQ_WAIT_ANY / SQ_WAVE_CYCLES)", c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]),
    ("wave_time: issue-stalled (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)", c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]),
    ("wave_time: of which stalled on LDS issue (SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES)", c["SQ_WAIT_INST_LDS"] / c["SQ_WAVE_CYCLES"]),
    ("VALU_insts_per_LDS_inst", c["SQ_INSTS_VALU"] / c["SQ_INSTS_LDS"]),
]
if "SQ_THREAD_CYCLES_VALU" in c:   # (round 6) both counters tick in the same unit: a kernel with every lane on reads 1.000 (raydirs_kernel)
    d.append(("VALU_lane_utilisation (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU))",
              c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])))
    d.append(("VALU_issue_ceiling_ms (SQ_INSTS_VALU x 4 cycles / 1024 SIMDs at 2.4 GHz: the kernel if nothing but its VALU "
              "instructions took time)", c["SQ_INSTS_VALU"] * 4 / 1024 / 2.4e6))
with open(out, "w") as f:
    w = csv.writer(f)
    w.writerow(["quantity", "value"])
    for k, v in rows:
        w.writerow([k, v])
    w.writerow(["--- raw counters, per launch ---", ""])
    for k in sorted(c):
        w.writerow([k, "%.6g" % c[k]])
    w.writerow(["--- derived ---", ""])
    for k, v in d:
        w.writerow([k, "%.4f" % v])
print(open(out).read())
