"""Summary of the rocprofv3 passes of tools/profile_mf.sh:  python tools/mf_summary.py TAG OUT_DIR CELLS  ->  profiles/TAG_mf_*"""
import csv, glob, json, re, sys
tag, out, cells = sys.argv[1], sys.argv[2], int(sys.argv[3])
st4 = glob.glob(out + "/stats4/**/*kernel_stats.csv", recursive=True)[0]
rows4 = list(csv.reader(open(st4)))
open(f"profiles/{tag}_mf_kernel_stats.csv", "w").write("\n".join(",".join('"%s"' % c for c in r) for r in rows4[:16]) + "\n")
st = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)[0]   # ONE stream: durations add up to the wall clock
rows = list(csv.reader(open(st)))
open(f"profiles/{tag}_mf_1stream_kernel_stats.csv", "w").write("\n".join(",".join('"%s"' % c for c in r) for r in rows[:16]) + "\n")
k = json.load(open(out + "/pmc_all.json"))["kernels"]
fam = lambda n: ("gemm_gather" if "true>" in n and "k_gemm_tile" in n else "gemm" if "k_gemm_tile" in n else "leaf_inverse" if "leaf_inverse" in n
                 else "front_fused" if "k_mf_front" in n else "build" if "k_mf_build" in n else "pad" if "k_mf_pad" in n
                 else "assembly" if ("k_assemble" in n or "k_c0" in n) else None)
agg = {}
for name, e in k.items():
    f = fam(name)
    if f is None:
        continue
    a = agg.setdefault(f, {})
    for c, v in e.items():
        if isinstance(v, dict):
            a[c] = a.get(c, 0.0) + v["sum"]
solves = 2 * cells  # tools/mf_check.py --time-only: one warm-up and one timed solve
tot_ns = sum(float(r[2]) for r in rows[1:] if fam(r[0]))
res = {"command": f"tools/profile_mf.sh {tag} {cells}: rocprofv3 (kernel trace; then --pmc SQ set, FETCH_SIZE, WRITE_SIZE in separate passes) -- "
                  f"python3 tools/mf_check.py --time-only {cells}  (3D elasticity, 16^3 micro cells, two solves of {cells} cells)",
       "cells_per_solve": cells, "kernel_time_ms_per_solve": tot_ns / 2e6,
       "note": "times, time shares and TFLOP/s are of the ONE-stream run (HOMMX_MF_STREAMS=1: kernel durations do not overlap and add up to the "
               "wall clock; TAG_mf_1stream_kernel_stats.csv); the default runs two to four pieces of a chunk side by side on as many streams, "
               "where durations overlap (TAG_mf_kernel_stats.csv); counters do not depend on the streams",
       "families": {}}
for f, a in agg.items():
    t_ns = sum(float(r[2]) for r in rows[1:] if fam(r[0]) == f)
    e = {"time_share": t_ns / tot_ns, "time_us_per_cell": t_ns / 1e3 / solves}
    if "SQ_INSTS_MFMA" in a:
        e["mfma_per_cell"] = a["SQ_INSTS_MFMA"] / solves
        e["mfma_flops_per_cell"] = a["SQ_INSTS_MFMA"] / solves * 2048.0
        e["tflops_one_stream"] = e["mfma_flops_per_cell"] / (e["time_us_per_cell"] * 1e-6) / 1e12 if e["time_us_per_cell"] > 0 else 0.0
        e["valu_per_cell"] = a.get("SQ_INSTS_VALU", 0.0) / solves
    if a.get("SQ_BUSY_CU_CYCLES"):
        e["mfma_busy_fraction_of_simd_cycles"] = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4 * a["SQ_BUSY_CU_CYCLES"])
        e["wait_fraction_of_wave_cycles"] = a.get("SQ_WAIT_INST_ANY", 0.0) / max(a.get("SQ_WAVE_CYCLES", 1.0), 1.0)
    if "FETCH_SIZE" in a:
        e["hbm_read_bytes_per_cell_x2_corrected"] = 2 * 1024 * a["FETCH_SIZE"] / solves
    if "WRITE_SIZE" in a:
        e["hbm_write_bytes_per_cell"] = 1024 * a["WRITE_SIZE"] / solves
    res["families"][f] = e
res["hbm_bytes_per_cell_total"] = sum(e.get("hbm_read_bytes_per_cell_x2_corrected", 0) + e.get("hbm_write_bytes_per_cell", 0) for e in res["families"].values())
res["mfma_flops_per_cell_total"] = sum(e.get("mfma_flops_per_cell", 0) for e in res["families"].values())
json.dump(res, open(f"profiles/{tag}_mf_pmc_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
