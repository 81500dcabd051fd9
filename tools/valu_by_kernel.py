#!/usr/bin/env python3
"""SQ_INSTS_VALU / SQ_INSTS_SALU per kernel of ONE bench step (rocprofv3 --pmc, one pass):
    python3 tools/valu_by_kernel.py plane [0|1]      (scan_bound)"""
import collections, csv, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
w = sys.argv[1]
bound = sys.argv[2] if len(sys.argv) > 2 else "0"
d = "/tmp/valu_by_kernel"
subprocess.run(["rm", "-rf", d])
cmd = ["rocprofv3", "--pmc", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES", "--kernel-trace", "-d", d, "--output-format",
       "csv", "--", "python3", "tools/scan_once.py", w, "3", bound]
subprocess.run(cmd, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for row in csv.DictReader(open(f)):
    name = row["Kernel_Name"].split("(")[0][-60:]
    acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "SQ_WAVES":
        cnt[name] += 1
tot = sum(v["SQ_INSTS_VALU"] for v in acc.values())
for name, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"])[:16]:
    print("%-62s launches %3d  VALU %8.2f M/step (%4.1f %%)  SALU %7.2f M/step" % (
        name, cnt[name], v["SQ_INSTS_VALU"] / 3e6, 100 * v["SQ_INSTS_VALU"] / tot, v["SQ_INSTS_SALU"] / 3e6))
print("total VALU per step: %.1f M  -> %.3f ms at 614.4 G wave-instr/s" % (tot / 3e6, tot / 3 / 614.4e9 * 1e3))
