"""tools/microbench <rows.json> writes one row per (instruction, waves per SIMD); this turns it into the summary
bench.py quotes beside the issue roof: python tools/microbench_summary.py rows.json profiles/r05_microbench.json r05"""
import json
import sys

rows = json.load(open(sys.argv[1]))
tag = sys.argv[3] if len(sys.argv) > 3 else "r05"
r8 = {r["op"]: r["nominal_cycles_per_wave_instruction"] for r in rows["rows"] if r["waves_per_simd"] == 8}
r2 = {r["op"]: r["nominal_cycles_per_wave_instruction"] for r in rows["rows"] if r["waves_per_simd"] == 2}
out = {"what": "tools/microbench.hip on an MI355X gpurun box (round %s): whole-chip issue rate of the vector instructions "
               "the scan kernels are made of; 256 x W workgroups of 256 threads (W waves on every SIMD), cycles = time x "
               "2.4 GHz x 1024 SIMDs / wave-instructions (nominal: the chip clocks below 2.4 GHz under load)" % tag,
       "cycles": {"v_fma_f32": r8.get("v_fma_f32"), "v_pk_fma_f32": r8.get("v_pk_fma_f32"),
                  "v_add_f64": r8.get("v_add_f64"), "v_fma_f64": r8.get("v_fma_f64"),
                  "v_min3_f32_abs": r8.get("v_min3_f32_abs"), "v_min_u32": r8.get("v_min_u32"),
                  "level2_mix": r8.get("level2_squares_14"),                       # the r04 / r05 level-2 body (squares)
                  "level2_mix_r03_abs_form": r8.get("level2_mix_20"),
                  "v_cmp_lt_f32_to_sgpr_plus_bcnt": r8.get("cmp32vcc+bcnt"),
                  "mfma_f64_16x16x4": r8.get("mfma_f64_16x16x4")},
       "cycles_at_2_waves_per_simd": {k: r2.get(k) for k in ("v_fma_f32", "v_pk_fma_f32", "level2_squares_14")},
       "rows": rows["rows"]}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["cycles"]))
