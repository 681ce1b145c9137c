"""Summarise rocprofv3 --pmc output (one or more *_counter_collection.csv) per kernel.

    python tools/pmc_summary.py OUT.json DIR_OR_CSV [DIR_OR_CSV ...] [--match substr]

For every kernel: number of dispatches and, per counter, the SUM over dispatches and the mean per dispatch.  Derived
figures follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB-ish units of 1 KB = 1024 B on
this tool chain's csv (reported as-is plus bytes = value * 1024), and on gfx950 FETCH_SIZE under-reports 16 B/lane
coalesced loads by 2x (hbm_read_bytes_corrected = 2 * bytes; see DESIGN.md section 6 for the cross-check).
"""
import collections, csv, glob, json, os, sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = None
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
        args.remove(match)
    out, srcs = args[0], args[1:]
    files = []
    for s in srcs:
        files += [s] if os.path.isfile(s) else glob.glob(os.path.join(s, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if match and match not in name:
                continue
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[(name, r["Counter_Name"])].add((f, r["Dispatch_Id"]))
    res = {}
    for name, ctrs in agg.items():
        e = {}
        for c, v in ctrs.items():
            n = len(disp[(name, c)])
            e[c] = {"sum": v, "dispatches": n, "per_dispatch": v / max(n, 1)}
        if "FETCH_SIZE" in e:
            e["hbm_read_bytes_per_dispatch_raw"] = e["FETCH_SIZE"]["per_dispatch"] * 1024
            e["hbm_read_bytes_per_dispatch_corrected_x2"] = 2 * e["hbm_read_bytes_per_dispatch_raw"]
        if "WRITE_SIZE" in e:
            e["hbm_write_bytes_per_dispatch"] = e["WRITE_SIZE"]["per_dispatch"] * 1024
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "SQ_BUSY_CU_CYCLES" in e:
            # MFMA_BUSY counts per SIMD, BUSY_CU per CU: 4 SIMDs per CU
            e["mfma_busy_fraction_of_simd_cycles"] = e["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / (4 * e["SQ_BUSY_CU_CYCLES"]["sum"])
        res[name] = e
    json.dump({"sources": [os.path.relpath(f) for f in files], "kernels": res}, open(out, "w"), indent=1)
    for name, e in sorted(res.items()):
        print(name[:80], {k: (round(v, 3) if isinstance(v, float) else round(v["per_dispatch"], 1)) for k, v in e.items()})


if __name__ == "__main__":
    main()
