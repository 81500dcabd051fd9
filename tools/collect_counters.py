#!/usr/bin/env python3
"""Collect the SQ instruction / activity counters and the HBM traffic of one workload's scan kernel on the GPU
box and write profiles-ready JSON (stamped with the content hash of the kernel sources, which bench.py checks
before quoting it):

    python3 tools/collect_counters.py plane [out_dir] [full_count|early_exit]   # -> out_dir/r03_plane_<rate>_scan_counters.json

Six separate rocprofv3 passes (counters only + --kernel-trace, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
WRITE_SIZE each in their own pass, gfx950 correction FETCH_SIZE x 2) of `tools/scan_once.py <workload>`, plus one
--kernel-trace --stats pass for the kernel's average duration."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402

# the kernels of the scan phase (bounded two-level scan: the level-1 passes and the balanced second level)
CELLS = ("k_cells_bounds", "k_scan_pairs", "k_scan_cells", "k_bound_axis")
KERNEL = {"plane": CELLS, "sphere": CELLS, "line": CELLS, "us": ("k_scan_us_f32", "k_scan_us_h16", "k_us_prep_h16", "k_us_recheck"),
          "dense": ("k_scan_dense_mfma", "k_scan_dense_h16", "k_dense_prep_h16", "k_dense_recheck"),
          "phantom": ("k_scan_us_f32", "k_scan_phantom_h16", "k_phantom_prep_h16", "k_us_recheck")}


def is_scan(w, name):
    return any(k in name for k in KERNEL[w])

LAUNCHES = 0
SETS = ["SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM",
        "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES",
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY",
        "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU",
        "FETCH_SIZE", "WRITE_SIZE"]
MFMA_SETS = ["SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES",
             "SQ_INSTS_VALU_MFMA_MOPS_F16"]   # (a pass whose counter the tool does not know fails and is skipped)


BOUND = "1"


def run_pass(w, counters, d):
    cmd = ["rocprofv3", "--pmc"] + counters.split() + ["--kernel-trace", "-d", d, "--output-format", "csv", "--",
                                                       "python3", "tools/scan_once.py", w, "1", BOUND]
    r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=300)
    if r.returncode != 0:
        return None
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        return None
    acc = collections.defaultdict(float)
    launches = collections.defaultdict(int)
    for row in csv.DictReader(open(f[0])):
        if is_scan(w, row["Kernel_Name"]):
            acc[row["Counter_Name"]] += float(row["Counter_Value"])
            launches[row["Counter_Name"]] += 1
    global LAUNCHES
    LAUNCHES = max(launches.values()) if launches else 0
    return dict(acc)   # SUM over the scan launches of the one bench step (bounded scan: pilots + second pass)


def main():
    global BOUND
    w = sys.argv[1]
    out_dir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out")
    mode = sys.argv[3] if len(sys.argv) > 3 else "early_exit"
    BOUND = "0" if mode == "full_count" else "1"
    os.makedirs(out_dir, exist_ok=True)
    os.environ.setdefault("TMPDIR", "/tmp")
    scratch = os.path.join(out_dir, "pmc_%s_%s" % (w, mode))
    c = {}
    for i, s in enumerate(SETS + (MFMA_SETS if w in ("dense", "us", "phantom") else [])):
        got = run_pass(w, s, os.path.join(scratch, "p%d" % i))
        if got is None:
            print("pass failed: " + s, file=sys.stderr)
            continue
        c.update(got)
        print("pass %d done: %s" % (i, s), flush=True)
    # kernel duration from a --kernel-trace --stats pass of the same command (3 launches)
    d = os.path.join(scratch, "stats")
    subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "-d", d, "--output-format", "csv", "--", "python3",
                    "tools/scan_once.py", w, "3", BOUND], cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                   timeout=300)
    avg_ns = None
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        tot = 0.0
        for row in csv.DictReader(open(f)):
            if is_scan(w, row["Name"]):
                tot += float(row["TotalDurationNs"])
        if tot:
            avg_ns = tot / 3.0        # per bench step (3 steps in this pass)
    out = {"workload": w, "rate": mode, "kernel": " + ".join(KERNEL[w]), "kernel_source_hash": kernel_source_hash(),
           "collected_at": time.strftime("%Y-%m-%d %H:%M:%S"),
           "source": "tools/collect_counters.py %s %s: rocprofv3 --pmc passes (one counter set each, --kernel-trace only) of "
                     "tools/scan_once.py %s 1 %s -- the bench's shapes and sampler stream, one step, scan_bound %s" % (
                         w, mode, w, BOUND, BOUND),
           "counters_per_launch_sum_over_chip": c, "scan_launches_per_step": LAUNCHES,
           "kernel_avg_ms": avg_ns / 1e6 if avg_ns else None,
           "note": "counters and kernel time are SUMS over the scan launches of ONE bench step (lsqr_batch_fit)"}
    if "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"] > 0:
        cyc = c["SQ_BUSY_CYCLES"] / 32.0            # per shader engine -> cycles of the launch
        out["cycles_per_simd"] = cyc
        if "SQ_INSTS_VALU" in c:
            out["valu_wave_instructions"] = c["SQ_INSTS_VALU"]
            out["valu_issue_busy"] = c["SQ_INSTS_VALU"] / 1024.0 * 4.0 / cyc
        if "SQ_INSTS_SALU" in c:
            out["salu_wave_instructions"] = c["SQ_INSTS_SALU"]
            out["salu_issue_busy_per_cu"] = c["SQ_INSTS_SALU"] / 256.0 / cyc
        if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_INSTS_VALU"):
            out["lanes_active"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_INSTS_VALU"] * 64.0)
        # matrix pipe (r05): busy cycles the SQ counted / SIMD-cycles of the launch, and the same from the instruction
        # count (32 cycles per v_mfma_f32_32x32x16_f16 / v_mfma_f64_16x16x4) -- both in shader-clock cycles, so neither
        # depends on an assumed clock; the clock the launch ran at follows from cycles and the kernel's duration
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            out["mfma_pipe_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
        if c.get("SQ_INSTS_MFMA"):
            out["mfma_wave_instructions"] = c["SQ_INSTS_MFMA"]
            out["mfma_pipe_busy_from_insts"] = c["SQ_INSTS_MFMA"] * 32.0 / (cyc * 1024.0)
        if avg_ns:
            out["shader_clock_ghz"] = cyc / avg_ns
    if "FETCH_SIZE" in c:
        rd = c["FETCH_SIZE"] * 1024 * 2             # gfx950: FETCH_SIZE under-reports wide streaming reads by 2
        wr = c.get("WRITE_SIZE", 0.0) * 1024
        out["hbm_read_bytes_corrected"] = rd
        out["hbm_write_bytes"] = wr
        out["hbm_bytes_per_launch"] = rd + wr
        out["hbm_note"] = "FETCH_SIZE_KB*1024*2 (gfx950 correction) + WRITE_SIZE_KB*1024, separate passes"
    path = os.path.join(out_dir, "%s_%s_%s_scan_counters.json" % (os.environ.get("LSQR_ROUND", "r05"), w, mode))
    json.dump(out, open(path, "w"), indent=1)
    print(path)
    subprocess.run(["rm", "-rf", scratch])


if __name__ == "__main__":
    main()
