"""Compile-time variants of the sweep kernels, built as whole libraries under build/var/ (git-ignored, shipped by
gpurun).  usage: python tools/experiments/ti1_variants.py ; then bash tools/experiments/ti1_variants_run.sh on the GPU box"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "poppunk_amd", "csrc")
OUT = os.path.join(ROOT, "build", "var")
os.makedirs(OUT, exist_ok=True)
base = open(os.path.join(SRC, "ppk_iterate.hip")).read()


def rep(text, a, b, count=1):
    assert text.count(a) >= 1, a
    return text.replace(a, b, count)


NT_A = "      if constexpr (!BY_OFFSET) d[k] = dist[rr];"
NT_B = "      if constexpr (!BY_OFFSET) { const f32x2 t_ = __builtin_nontemporal_load(reinterpret_cast<const f32x2 *>(dist) + rr); d[k] = make_float2(t_.x, t_.y); }"
variants = {
    0: base,
    1: rep(base, "constexpr int kExpandBatch = 8;", "constexpr int kExpandBatch = 16;"),
    2: rep(base, NT_A, NT_B),
    3: rep(rep(base, NT_A, NT_B), "constexpr int kExpandBatch = 8;", "constexpr int kExpandBatch = 16;"),
    4: rep(base, "constexpr int kEmitBlock = 1024;", "constexpr int kEmitBlock = 256;"),
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
