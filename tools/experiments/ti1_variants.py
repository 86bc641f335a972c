"""Variants of the sweep's classify pass, built as whole libraries under build/var/ (git-ignored, shipped by gpurun):
occupancy bound, rows per dense batch, rows per lane.  usage: python tools/experiments/ti1_variants.py
then  bash tools/experiments/ti1_variants_run.sh  (on the GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "poppunk_amd", "csrc")
OUT = os.path.join(ROOT, "build", "var")
os.makedirs(OUT, exist_ok=True)
base = open(os.path.join(SRC, "ppk_iterate.hip")).read()
LB = "template <int MODE, bool FILTER, typename F>\n__global__ void __launch_bounds__(256)\nti1_classify_kernel("
assert base.count(LB) == 1


def lb(text, n):
    return text.replace(LB, LB.replace("__launch_bounds__(256)", "__launch_bounds__(256, %d)" % n))


def dense(text, n):
    return text.replace("constexpr int kDenseRows = 2;", "constexpr int kDenseRows = %d;" % n)


def unit(text, n):
    return text.replace("constexpr int kUnitWords = 8;", "constexpr int kUnitWords = %d;" % n)


variants = {
    0: base,
    1: lb(base, 5),
    2: lb(base, 6),
    3: lb(dense(base, 1), 6),
    4: lb(dense(base, 4), 5),
    5: lb(unit(base, 4), 6),
    6: lb(unit(dense(base, 1), 4), 8),
}
objs = [os.path.join(SRC, f) for f in ("ppk_api.o", "ppk_host.o", "ppk_dist.o", "ppk_boundary.o", "ppk_square.o", "ppk_sparse.o", "ppk_h5.o")]
procs = []
for v, text in variants.items():
    f = os.path.join(OUT, "ppk_iterate_v%d.hip" % v)
    open(f, "w").write(text)
    o = f.replace(".hip", ".o")
    procs.append((v, o, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
                    "-I" + SRC, "-I" + os.path.join(ROOT, "include"), "-c", f, "-o", o], stderr=subprocess.DEVNULL)))
for v, o, pr in procs:
    assert pr.wait() == 0, v
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, "libppk_v%d.so" % v), o] + objs + ["-ldl"], check=True)
    print("built variant", v)
