#!/bin/bash
# rocprofv3 evidence for the small-block kernels (run on the GPU box from the repo root):  tools/profile_small.sh TAG
#   pass 1: --kernel-trace --stats                                     -> profiles/TAG_small_kernel_stats.csv
#   pass 2-3: --pmc SQ sets (counters only, never mixed with tracing)   -> profiles/TAG_small_pmc_summary.json
set -e
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out/prof_small_$TAG
rm -rf $OUT && mkdir -p $OUT profiles
CMD="python3 tools/bench_kinds.py --small"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/sq -o run -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $OUT/lds -o run -- $CMD > $OUT/lds.log 2>&1
python3 tools/pmc_summary.py $OUT/pmc_all.json $OUT/sq $OUT/lds --match k_small
python3 - "$TAG" "$OUT" <<'PY'
import csv, glob, json, sys
tag, out = sys.argv[1], sys.argv[2]
st = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.reader(open(st)))
open(f"profiles/{tag}_small_kernel_stats.csv", "w").write("\n".join(",".join('"%s"' % c for c in r) for r in rows[:12]) + "\n")
k = json.load(open(out + "/pmc_all.json"))["kernels"]
avg = {r[0].split("(")[0].replace("void ", ""): float(r[3]) for r in rows[1:]}
cells = {"k_small_fused<64, 1, 9, 4>": 4096}
res = {"command": "tools/profile_small.sh: rocprofv3 --pmc <two SQ sets, separate passes> -- python3 tools/bench_kinds.py --small (3 launches per kernel)", "kernels": {}}
for name, e in k.items():
    nc = cells.get(name.replace("hommx::", ""), 8192)
    per = lambda c: e[c]["per_dispatch"] if c in e else None
    res["kernels"][name] = {
        "cells_per_launch": nc,
        "kernel_avg_ns_kernel_trace": avg.get(name),
        "valu_per_cell": per("SQ_INSTS_VALU") / nc, "mfma_per_cell": per("SQ_INSTS_MFMA") / nc,
        "salu_per_cell": per("SQ_INSTS_SALU") / nc, "lds_insts_per_cell": per("SQ_INSTS_LDS") / nc,
        "mfma_busy_fraction_of_simd_cycles": e.get("mfma_busy_fraction_of_simd_cycles"),
        "lds_bank_conflict_per_launch": per("SQ_LDS_BANK_CONFLICT"), "lds_idx_active_per_launch": per("SQ_LDS_IDX_ACTIVE"),
        "wave_cycles_per_launch": per("SQ_WAVE_CYCLES"), "busy_cu_cycles_per_launch": per("SQ_BUSY_CU_CYCLES"),
    }
json.dump(res, open(f"profiles/{tag}_small_pmc_summary.json", "w"), indent=1)
print(json.dumps(res["kernels"], indent=1))
PY
