"""Dev tool: per-dispatch table of the LAST solve in a rocprofv3 kernel trace (csv): python tools/trace_table.py DIR [min_us]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
idx = [i for i, r in enumerate(rows) if "k_assemble" in r["Kernel_Name"]]
rows = rows[idx[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
tot = 0.0
for r in rows:
    n = r["Kernel_Name"].replace("void ", "").replace("hommx::", "").split("(")[0]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += dur
    if dur >= (float(sys.argv[2]) if len(sys.argv) > 2 else 0.0):
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} us  {n[:60]:60s} wg {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):8d} x {r['Workgroup_Size_X']:>4s}  {dur:9.1f} us  vgpr {r['VGPR_Count']} agpr {r['Accum_VGPR_Count']} lds {r['LDS_Block_Size']}")
print(f"sum of kernel durations {tot:.1f} us; span {(int(rows[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
