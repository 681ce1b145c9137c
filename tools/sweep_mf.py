"""Dev tool: sweep of environment knobs of the nested-dissection route at the C4 / C5 size (each setting in a child process, best of 3)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cells = sys.argv[1] if len(sys.argv) > 1 else "512"
settings = [{}] + [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[2:]]
for env in settings:
    best = 0.0
    for _ in range(3):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mf_check.py"), "--time-only", cells], env=dict(os.environ, **env),
                           capture_output=True, text=True)
        m = re.search(r"([\d.]+) solves/s", r.stdout)
        best = max(best, float(m.group(1)) if m else 0.0)
    print(f"{str(env):60s} best of 3: {best:8.1f} solves/s", flush=True)
