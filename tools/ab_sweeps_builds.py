"""Same-box comparison of two builds of the library on the boundary sweeps (tools/time_sweeps.py's workload):
    python tools/ab_sweeps_builds.py old.so new.so [rounds]
Each build runs in its own process, alternating; an older build lacks entry points added since, which are left unbound."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, ctypes, runpy
sys.path.insert(0, %r)
from poppunk_amd import _lib
_lib.SO_PATH = os.path.abspath(sys.argv[1])
_lib._preload_hip_runtime()
_h = ctypes.CDLL(_lib.SO_PATH)
for _n in list(_lib.SIGNATURES):
    if not hasattr(_h, _n):
        del _lib.SIGNATURES[_n]
sys.argv = ["time_sweeps.py", "--plain", "--check"]
runpy.run_path(os.path.join(%r, "tools", "time_sweeps.py"), run_name="__main__")
''' % (ROOT, ROOT)
libs = [a for a in sys.argv[1:] if not a.isdigit()]
rounds = int(([a for a in sys.argv[1:] if a.isdigit()] or ["3"])[0])
for r in range(rounds):
    for lib in libs:
        out = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True)
        for line in out.stdout.splitlines():
            if "median" in line or "equals" in line:
                print("%-28s %s" % (os.path.basename(lib), line.strip()), flush=True)
        if out.returncode:
            print(os.path.basename(lib), "failed:", out.stderr[-400:])
