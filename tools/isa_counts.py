#!/usr/bin/env python3
"""Vector / scalar / memory instructions per loop of the scan kernels, counted from the ISA of THIS tree
(lsqrrecipes_amd/csrc/lsqr_hip.s: `make -C lsqrrecipes_amd/csrc asm`), so that the `useful` instruction counts bench.py
prices the issue roof with (SCAN_USEFUL, counted from the source) can be read next to what the compiler emitted
(VERDICT r04 item 4).  Runs on the CPU (hipcc cross-compiles); writes profiles/r05_isa_counts.json, stamped with the
kernel source hash bench.py checks.

    python tools/isa_counts.py [out.json]"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402

CSRC = os.path.join(ROOT, "lsqrrecipes_amd", "csrc")
# (key, substrings the mangled name must contain)
KERNELS = [("k_scan_pairs_plane", ["k_scan_pairs", "PlaneCell", "Li3E"]),
           ("k_scan_pairs_sphere", ["k_scan_pairs", "SphereCell", "Li3E"]),
           ("k_scan_pairs_line", ["k_scan_pairs", "LineCell", "Li3E"]),
           ("k_scan_us_f32", ["k_scan_us_f32", "USModel", "Lb1E"]),
           ("k_scan_dense_h16", ["k_scan_dense_h16"]),
           ("k_scan_us_h16", ["k_scan_us_h16", "Lb1E"]),
           ("k_scan_phantom_h16", ["k_scan_phantom_h16"]),
           ("k_lm_persist_us", ["k_lm_persist", "USModel", "Lb1E", "Li0E"])]


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_pk_"):
        return "valu_packed"
    if op.startswith("v_cmp"):
        return "valu_cmp"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "control"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_isa_counts.json")
    asm = os.path.join(CSRC, "lsqr_hip.s")
    if not os.path.exists(asm) or os.path.getmtime(asm) < max(
            os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))):
        subprocess.check_call(["make", "-C", CSRC, "asm"])
    text = open(asm).read()
    # function bodies: "<mangled>:" ... ".Lfunc_end"
    starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z[\w]+):\s*(?:;.*)?$", text, re.M)]
    res = {}
    for key, subs in KERNELS:
        cand = [(p, n) for p, n in starts if all(s_ in n for s_ in subs)]
        if not cand:
            continue
        # the largest instantiation (the default cell size / the production variant is the biggest body)
        best = None
        for p, n in cand:
            e = text.index(".Lfunc_end", p)
            if best is None or e - p > best[2] - best[0]:
                best = (p, n, e)
        body = text[best[0]:best[2]].split("\n")
        loops = collections.OrderedDict()
        cur = ("entry", 0)
        for li, line in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):\s*;?\s*(.*)$", line)
            if m:
                lab, rest = m.group(1), m.group(2)
                # a nested loop's header carries its parents on the label line and "This (Inner) Loop Header: Depth=n"
                # on one of the comment lines that follow
                k = li + 1
                while "Parent Loop" in rest and k < len(body) and body[k].lstrip().startswith(";"):
                    if "Loop Header" in body[k]:
                        rest = body[k]
                        break
                    k += 1
                d = re.search(r"Depth=(\d+)", rest)
                h = re.search(r"Header=(BB\d+_\d+)", rest)
                if "Loop Header" in rest and d:
                    cur = (lab.lstrip("."), int(d.group(1)))
                elif h and d:
                    cur = (h.group(1), int(d.group(1)))
                else:
                    cur = ("straight", 0)
                continue
            if not line.startswith("\t") or line.startswith("\t.") or line.startswith("\t;"):
                continue
            op = line.split()[0]
            loops.setdefault(cur, collections.Counter())[classify(op)] += 1
            loops[cur]["op:" + op] += 1
        rows = []
        for (hdr, depth), cnt in loops.items():
            if depth == 0:
                continue
            valu = cnt["valu_packed"] + cnt["valu_cmp"] + cnt["valu_other"]
            top = {k[3:]: v for k, v in cnt.items() if k.startswith("op:") and k.startswith("op:v_")}
            rows.append({"loop": hdr, "depth": depth, "valu": valu, "valu_packed": cnt["valu_packed"],
                         "valu_cmp": cnt["valu_cmp"], "mfma": cnt["mfma"], "salu": cnt["salu"], "smem": cnt["smem"],
                         "lds": cnt["lds"], "vmem": cnt["vmem"], "control": cnt["control"],
                         "vector_ops": dict(sorted(top.items(), key=lambda kv: -kv[1])[:14])})
        rows.sort(key=lambda r: (-r["depth"], -r["valu"]))
        # the loop with the most packed instructions = the packed fp32 filter's body (level 2 of the point models)
        l2 = max(rows, key=lambda r: r["valu_packed"]) if rows else None
        res[key] = {"function": best[1][:120], "loops": rows,
                    "packed_body": {"loop": l2["loop"], "valu": l2["valu"], "valu_packed": l2["valu_packed"],
                                    "valu_cmp": l2["valu_cmp"], "salu": l2["salu"]} if l2 else None}
    out = {"what": "tools/isa_counts.py: instructions per loop of the scan kernels in this tree's ISA (one pass over a loop "
                   "body; a body the compiler unrolled covers several source iterations)",
           "kernel_source_hash": kernel_source_hash(), "kernels": res}
    json.dump(out, open(out_path, "w"), indent=1)
    for k, v in res.items():
        print(k, v["packed_body"])


if __name__ == "__main__":
    main()
