#!/bin/bash
# rocprofv3 evidence for the bench line (run on the GPU box from the repo root):  tools/profile_c2.sh TAG
#   pass 1: --kernel-trace --stats (average kernel duration)          -> profiles/TAG_bench_kernel_stats.csv
#   pass 2-4: --pmc FETCH_SIZE | WRITE_SIZE | SQ set (separate passes, counters only: never mixed with tracing)
#   summary: profiles/TAG_c2_pmc_summary.json (HBM bytes per launch: 2 x FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes
#   for 16 B/lane reads on gfx950; VALU / MFMA / LDS instruction counts per launch)
set -e
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT profiles
CMD="python3 bench.py --no-c5 --no-cpu-baseline --no-host-boundary --steps 20 --warmup 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o run -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/write -o run -- $CMD > $OUT/write.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/sq -o run -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $OUT/lds -o run -- $CMD > $OUT/lds.log 2>&1
python3 tools/pmc_summary.py $OUT/pmc_all.json $OUT/fetch $OUT/write $OUT/sq $OUT/lds --match k_poisson2d_fused
python3 - "$TAG" "$OUT" <<'PY'
import csv, glob, json, sys
tag, out = sys.argv[1], sys.argv[2]
st = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.reader(open(st)))
open(f"profiles/{tag}_bench_kernel_stats.csv", "w").write("\n".join(",".join('"%s"' % c for c in r) for r in rows[:6]) + "\n")
k = json.load(open(out + "/pmc_all.json"))["kernels"]
name = [n for n in k if "k_poisson2d_fused<32>" in n][0]
e = k[name]
per = lambda c: e[c]["per_dispatch"]
fetch, write = per("FETCH_SIZE") * 1024, per("WRITE_SIZE") * 1024
avg_ns = [float(r[3]) for r in rows[1:] if "k_poisson2d_fused<32>" in r[0]][0]
s = {
    "command": "tools/profile_c2.sh: rocprofv3 --pmc <counters> -- python3 bench.py --no-c5 --no-cpu-baseline --no-host-boundary --steps 20 --warmup 3 (separate passes: FETCH_SIZE; WRITE_SIZE; two SQ sets), summarised by tools/pmc_summary.py",
    "kernel": "k_poisson2d_fused<32>",
    "cells_per_launch": 8192,
    "kernel_avg_ns_kernel_trace": avg_ns,
    "FETCH_SIZE_KB_per_launch": per("FETCH_SIZE"),
    "WRITE_SIZE_KB_per_launch": per("WRITE_SIZE"),
    "fetch_bytes_corrected_x2": 2 * fetch,
    "write_bytes": write,
    "hbm_bytes_per_launch": 2 * fetch + write,
    "algorithmic_bytes_per_launch": 8192 * 16416,
    "note": "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide (16 B/lane) coalesced reads on gfx950",
    "sq": {c: per(c) for c in e if c.startswith("SQ_")},
}
s["valu_per_solve"] = s["sq"]["SQ_INSTS_VALU"] / 8192
s["mfma_per_solve"] = s["sq"]["SQ_INSTS_MFMA"] / 8192
if "SQ_LDS_IDX_ACTIVE" in s["sq"]:
    s["lds_utilisation"] = s["sq"]["SQ_LDS_IDX_ACTIVE"] / s["sq"]["SQ_BUSY_CU_CYCLES"]
json.dump(s, open(f"profiles/{tag}_c2_pmc_summary.json", "w"), indent=1)
print(json.dumps(s, indent=1))
PY
cp $OUT/stats.log gpurun_out/${TAG}_stats.log 2>/dev/null || true
