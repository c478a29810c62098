#!/usr/bin/env python3
"""Per-kernel resource usage (VGPR / AGPR / SGPR / LDS / scratch) of the gfx950 code objects inside .o files:
   python scripts/dev/kres.py neural_svd_amd/csrc/build/pmlp_bwd.o [name-filter]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(obj):
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True, stdout=subprocess.DEVNULL, cwd=tmp)
        for co in sorted(f for f in os.listdir(tmp) if "amdgcn" in f):
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(tmp, co)], check=True,
                                   capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                blk = ".agpr_count:" + blk
                g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
                name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
                yield name, dict(vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"),
                                 lds=g("group_segment_fixed_size"), scratch=g("private_segment_fixed_size"),
                                 spill_v=g("vgpr_spill_count"), spill_s=g("sgpr_spill_count"))


if __name__ == "__main__":
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, r in kernels(sys.argv[1]):
        if flt in name:
            short = re.sub(r"^void ", "", name)
            short = re.sub(r"\((anonymous namespace::)?[A-Z]\w*Args.*", "", short)[:70]
            print(f"{short:70s} " + " ".join(f"{k}={v}" for k, v in r.items()))
