"""Dev tool: knob sweep on one bench_kinds case:  python tools/sweep_kinds.py dim,n,kind,cells K=V[,K=V] ..."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
case = sys.argv[1]
for env in [{}] + [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[2:]]:
    best = 0.0
    for _ in range(2):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_kinds.py"), f"--case={case}"], env=dict(os.environ, **env),
                           capture_output=True, text=True)
        m = re.search(r"([\d.]+) solves/s", r.stdout)
        best = max(best, float(m.group(1)) if m else 0.0)
    print(f"{case:24s} {str(env):40s} {best:10.1f} solves/s", flush=True)
