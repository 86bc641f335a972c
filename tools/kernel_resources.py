"""Register / spill / LDS table of the dist_kernel_v2 instantiations (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py [extra hipcc flags]

Columns: MODE (0 dist, 1 jaccard, 2 counts, 3 mask, 4 knn), W, KSPLIT, WIDE, EXP."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "poppunk_amd", "csrc", "ppk_dist.hip")
with tempfile.TemporaryDirectory() as td:
    o = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                        "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(td, "x.o")] + sys.argv[1:],
                       capture_output=True, text=True, cwd=os.path.dirname(src))
rows, cur = [], None
for l in o.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark: +(.*?): (\S+) \[", l)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
print("%-22s %6s %6s %8s %8s %8s %6s" % ("MODE,W,KSPLIT,WIDE,EXP", "SGPR", "VGPR", "scratch", "s-spill", "v-spill", "LDS"))
for r in rows:
    m = re.search(r"dist_kernel_v2ILi8ELi(\d)ELi(\d)ELb(\d)ELb(\d)ELb(\d)", r["name"])
    if m:
        print("%-22s %6s %6s %8s %8s %8s %6s" % (",".join(m.groups()), r.get("TotalSGPRs"), r.get("VGPRs"),
              r.get("ScratchSize [bytes/lane]"), r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("LDS Size [bytes/block]")))
