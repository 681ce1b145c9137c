"""Dev tool: per-kernel time table from a rocprofv3 results database (rocpd sqlite, the default output format of ROCm 7.2):
   python tools/rocpd_stats.py gpurun_out/prof_x/x_results.db [--by-grid] [--csv out.csv]"""
import re, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
by_grid = "--by-grid" in sys.argv
rows = cur.execute(f"select s.kernel_name, d.end - d.start, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id = s.id").fetchall()
agg = {}
for name, ns, gx, wx in rows:
    name = re.sub(r"\(.*\)$", "", name.replace("hommx::", "").replace("void ", "").replace(" [clone .kd]", ""))
    key = (name, gx // max(1, wx)) if by_grid else name
    a = agg.setdefault(key, [0, 0])
    a[0] += 1
    a[1] += ns
tot = sum(v[1] for v in agg.values())
lines = ["name,calls,total_ms,avg_us,percent"]
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"\"{k}\",{n},{t/1e6:.3f},{t/n/1e3:.1f},{100*t/tot:.2f}")
print("\n".join(lines[:45]))
print(f"total kernel time {tot/1e6:.2f} ms over {len(rows)} dispatches")
if "--csv" in sys.argv:
    open(sys.argv[sys.argv.index("--csv") + 1], "w").write("\n".join(lines) + "\n")
