"""Per-GROUP table of the nested-dissection route from rocprofv3 output (tools/profile_mf.sh): which tree level costs what.

    python tools/mf_groups.py PROF_DIR PIECES [--json out.json]

PROF_DIR holds the passes of tools/profile_mf.sh (stats/ = kernel trace, fetch/, write/, sq/ = counter passes).  The route launches group
after group (multifrontal.hip: mf_solve), and inside a group piece after piece, every piece starting with k_mf_build -- or consisting of ONE k_mf_front launch (groups on the
register-resident front kernel): the dispatches between the (PIECES * g)-th and the (PIECES * (g + 1))-th such launch of a solve belong to group g
(leaves first).  Counter passes
serialise the kernels, so the dispatch order is the launch order; the kernel trace of a run on ONE stream (HOMMX_MF_STREAMS=1) gives
per-group times that add up to the wall clock.  HBM bytes: 2 x FETCH_SIZE + WRITE_SIZE in units of 1 KiB (the x 2 as
MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950; uncalibrated for 8-byte gathers -- ratios between levels hold).
"""
import csv, glob, json, os, sys


def short(name):
    n = name.replace("void ", "").replace("hommx::", "").split("(")[0]
    if "k_gemm_tile" in n:
        return "gemm_gather" if n.rstrip(">").endswith("true") else "gemm"
    for k in ("k_mf_build", "k_mf_pad", "k_leaf_inverse", "k_mf_front", "k_mf_finalize", "k_assemble", "k_c0", "k_expand"):
        if k in n:
            return k
    return n[:40]


def dispatches(csv_path, value_col=None):
    """[(dispatch_id, kernel short name, value)] in dispatch order; kernel trace: value = duration in ns."""
    rows = {}
    for r in csv.DictReader(open(csv_path)):
        d = int(r["Dispatch_Id"])
        if value_col is None:
            rows[d] = (short(r["Kernel_Name"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        elif r["Counter_Name"] == value_col:
            rows[d] = (short(r["Kernel_Name"]), rows.get(d, ("", 0.0))[1] + float(r["Counter_Value"]))
    return [(d,) + rows[d] for d in sorted(rows)]


def by_group(disp, pieces):
    """Split the dispatch list at every `pieces`-th k_mf_build; returns a list of solves, each a list of groups {kernel: sum}."""
    solves, cur, builds = [], None, 0
    for _, k, v in disp:
        if k in ("k_assemble", "k_expand"):  # a new solve (or a new chunk) starts with K1
            if cur is not None and builds:
                solves.append(cur)
                cur, builds = None, 0
        if k in ("k_mf_build", "k_mf_front"):  # every group of a piece starts with its build kernel or IS one front-kernel launch
            if cur is None:
                cur = []
            if builds % pieces == 0:
                cur.append({})
            builds += 1
        if cur:
            cur[-1][k] = cur[-1].get(k, 0.0) + v
    if cur:
        solves.append(cur)
    return solves


def main():
    d, pieces = sys.argv[1], int(sys.argv[2])
    find = lambda sub, pat: (glob.glob(os.path.join(d, sub, "**", pat), recursive=True) or [None])[0]
    out = {}
    tr = find("stats", "*kernel_trace.csv")
    if tr:
        out["time_ns"] = by_group(dispatches(tr), pieces)
    for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"), ("sq", "SQ_INSTS_MFMA")):
        f = find(sub, "*counter_collection.csv")
        if f:
            out[ctr] = by_group(dispatches(f, ctr), pieces)
    cells = int(sys.argv[sys.argv.index("--cells") + 1]) if "--cells" in sys.argv else 1
    table = []
    ng = max(len(s[-1]) for s in out.values() if s)
    for g in range(ng):
        row = {"group": g}
        last = lambda key: (out[key][-1][g] if key in out and out[key] and g < len(out[key][-1]) else {})
        t, fe, wr, mf = last("time_ns"), last("FETCH_SIZE"), last("WRITE_SIZE"), last("SQ_INSTS_MFMA")
        row["time_us_per_cell"] = {k: v / 1e3 / cells for k, v in t.items()}
        row["time_us_per_cell_total"] = sum(t.values()) / 1e3 / cells
        row["hbm_read_MB_per_cell_x2"] = {k: 2 * 1024 * v / 1e6 / cells for k, v in fe.items()}
        row["hbm_write_MB_per_cell"] = {k: 1024 * v / 1e6 / cells for k, v in wr.items()}
        row["hbm_MB_per_cell_total"] = (2 * 1024 * sum(fe.values()) + 1024 * sum(wr.values())) / 1e6 / cells
        row["mfma_gflop_per_cell"] = sum(mf.values()) * 2048 / 1e9 / cells
        if row["time_us_per_cell_total"] > 0:
            row["tflops"] = row["mfma_gflop_per_cell"] * 1e9 / (row["time_us_per_cell_total"] * 1e-6) / 1e12
            row["hbm_TBps"] = row["hbm_MB_per_cell_total"] * 1e6 / (row["time_us_per_cell_total"] * 1e-6) / 1e12
        table.append(row)
    for r in table:
        print(f"group {r['group']}: {r['time_us_per_cell_total']:8.1f} us/cell  {r['hbm_MB_per_cell_total']:8.1f} MB/cell  {r['mfma_gflop_per_cell']:6.2f} GFLOP/cell"
              + (f"  {r.get('tflops', 0):5.1f} TF/s  {r.get('hbm_TBps', 0):4.2f} TB/s" if "tflops" in r else ""))
        for k in sorted(set(r["time_us_per_cell"]) | set(r["hbm_read_MB_per_cell_x2"])):
            print(f"      {k:16s} {r['time_us_per_cell'].get(k, 0):8.1f} us   read {r['hbm_read_MB_per_cell_x2'].get(k, 0):7.1f}  write {r['hbm_write_MB_per_cell'].get(k, 0):7.1f} MB")
    print(f"total: {sum(r['time_us_per_cell_total'] for r in table):.1f} us/cell  {sum(r['hbm_MB_per_cell_total'] for r in table):.1f} MB/cell")
    if "--json" in sys.argv:
        json.dump(table, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
