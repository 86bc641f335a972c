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


CFG = "  typedef rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0> Cfg;"


def cfg(bs, ipt, algo="match", hist=(256, 12)):
    return rep(base, CFG, "  typedef rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, "
               "rocprim::radix_sort_onesweep_config<rocprim::kernel_config<%d, %d>, rocprim::kernel_config<%d, %d>, 8, "
               "rocprim::block_radix_rank_algorithm::%s>, 0> Cfg;" % (hist[0], hist[1], bs, ipt, algo))


variants = {
    0: cfg(1024, 8, hist=(1024, 16)),
    1: cfg(1024, 6, hist=(1024, 16)),
    2: cfg(1024, 4, hist=(1024, 16)),
    3: cfg(512, 8, hist=(1024, 16)),
    4: cfg(1024, 8, hist=(1024, 8)),
    5: cfg(1024, 8, hist=(512, 16)),
    6: cfg(1024, 10, hist=(1024, 16)),
    7: cfg(1024, 8, hist=(1024, 32)),
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
