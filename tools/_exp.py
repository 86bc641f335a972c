"""The tools that switch parts of the kernels off (`ablate`), pick a rejected tile order (`map`), run without strip
tiles (`strip`) or without the kept edge buffer (`edge_list_keep`) need the EXPERIMENTS build of the library:

    make -C poppunk_amd/csrc experiments        ->  poppunk_amd/csrc/libppk_hip_exp.so

`use()` points poppunk_amd._lib at it (PPK_LIBRARY) -- call it before importing poppunk_amd.  The product library
has no option of these names."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "poppunk_amd", "csrc", "libppk_hip_exp.so")


def use():
    if not os.path.exists(EXP):
        sys.exit("experiments library missing: make -C poppunk_amd/csrc experiments")
    os.environ["PPK_LIBRARY"] = EXP
