"""Dev tool: A/B of the register-resident front kernel (HOMMX_MF_FRONT: most tiles per dimension, 0 = off) on the nested-dissection route.
Each setting runs in a child process (the knob is read when a plan is created).  python tools/bench_mf_ab.py [--cases ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cases = ["2,128,poisson,2048", "2,96,poisson,2048", "2,64,elasticity,2048", "3,16,poisson,1024", "3,12,poisson,1024", "3,8,elasticity,1024",
         "3,12,elasticity,1024", "3,16,elasticity,512"]
if "--cases" in sys.argv:
    cases = sys.argv[sys.argv.index("--cases") + 1:]
settings = [("off", {"HOMMX_MF_FRONT": "0"}), ("default", {})]
for extra in [a for a in sys.argv[1:] if a.startswith("--env=")]:
    k, v = extra[6:].split("=", 1)
    settings.append((f"{k}={v}", {k: v}))
for c in cases:
    for tag, env in settings:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_kinds.py"), f"--case={c}"], env=dict(os.environ, **env),
                           capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if "solves/s" in ln]
        print(f"[{tag:>10s}] " + (line[-1] if line else "FAILED: " + r.stderr[-300:]), flush=True)
