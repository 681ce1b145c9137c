#!/bin/bash
# rocprofv3 evidence for BASELINE config 3 ("rocprof HBM GB/s reported"): PoissonStratifiedHMM, 128x128 macro, 32x32 micro, wavy laminate,
# on the fused 2D kernel (run on the GPU box from the repo root):  tools/profile_c3.sh TAG
#   pass 1: --kernel-trace --stats; pass 2-3: --pmc FETCH_SIZE | WRITE_SIZE (separate passes)  -> profiles/TAG_c3_pmc_summary.json
set -e
TAG=${1:-r04}
export TMPDIR=/tmp
OUT=gpurun_out/prof_c3_$TAG
rm -rf $OUT && mkdir -p $OUT profiles
CMD="python3 tools/run_c3.py 5"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o run -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/write -o run -- $CMD > $OUT/write.log 2>&1
python3 tools/pmc_summary.py $OUT/pmc_all.json $OUT/fetch $OUT/write --match k_poisson2d_fused > $OUT/pmc_summary.log
python3 - "$TAG" "$OUT" <<'PY'
import csv, glob, json, sys
tag, out = sys.argv[1], sys.argv[2]
st = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.reader(open(st)))
k = json.load(open(out + "/pmc_all.json"))["kernels"]
name = [n for n in k if "k_poisson2d_fused" in n][0]
e = k[name]
per = lambda c: e[c]["per_dispatch"]
avg_ns = [float(r[3]) for r in rows[1:] if "k_poisson2d_fused" in r[0]][0]
cells = 32768
alg = cells * 16448
hbm = 2 * per("FETCH_SIZE") * 1024 + per("WRITE_SIZE") * 1024
s = {"command": "tools/profile_c3.sh: rocprofv3 (--kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE, separate passes) -- python3 tools/run_c3.py 5",
     "config": "C3: PoissonStratifiedHMM, 128x128 macro (32,768 cells), 32x32 micro, wavy laminate, M = Dtheta^T(c_T) per cell",
     "kernel": name, "cells_per_launch": cells, "kernel_avg_ns_kernel_trace": avg_ns,
     "FETCH_SIZE_KB_per_launch": per("FETCH_SIZE"), "WRITE_SIZE_KB_per_launch": per("WRITE_SIZE"),
     "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": hbm / alg,
     "hbm_GBps": hbm / (avg_ns * 1e-9) / 1e9, "hbm_frac_of_8TBps": hbm / (avg_ns * 1e-9) / 8e12,
     "solves_per_s_kernel_trace": cells / (avg_ns * 1e-9),
     "note": "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide (16 B/lane) coalesced reads on gfx950; the path is compute-bound "
             "(375 flop per byte), so the HBM fraction is small by construction"}
json.dump(s, open(f"profiles/{tag}_c3_pmc_summary.json", "w"), indent=1)
print(json.dumps(s, indent=1))
PY
