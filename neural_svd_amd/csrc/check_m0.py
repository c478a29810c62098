#!/usr/bin/env python3
"""Build-time ISA check for the kernels that include tile128_dma.h.

t128d_dma() writes M0 from inline assembly (`s_mov_b32 m0, sN ; s_nop 0 ; global_load_lds_dwordx4`) without being able
to tell the compiler (M0 is reserved: hipcc rejects it on a clobber list). That is only sound while the compiler emits
no M0 use of its own in those kernels (movrel register-array indexing, the LDS-DMA builtin, ds_*_addtid, s_sendmsg with
an M0 payload ...). This script disassembles the code object of the build's architecture (NSVD_ARCH, default gfx950) inside each given .o and fails the build unless

  * every instruction that mentions m0 is `s_mov_b32 m0, <sgpr>`, and
  * each of them is followed by `s_nop 0` and then `global_load_lds_dwordx4` (or `_dword`: gemm16.h's bias pieces), and
  * every `global_load_lds_*` is preceded by exactly that pair.

Usage: check_m0.py build/pmlp_bwd.o build/tower.o
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
ARCH = os.environ.get("NSVD_ARCH", "gfx950")  # the Makefile passes its ARCH


def device_isa(obj):
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, stdout=subprocess.DEVNULL, cwd=tmp)
        cos = [f for f in os.listdir(tmp) if "amdgcn" in f and ARCH in f]
        if not cos:
            raise SystemExit(f"check_m0: no {ARCH} code object in {obj}")
        out = []
        for co in sorted(cos):
            out.append(subprocess.run([OBJDUMP, "-d", os.path.join(tmp, co)], check=True, capture_output=True,
                                      text=True).stdout)
        return "\n".join(out)


def instructions(isa):
    for line in isa.splitlines():
        line = line.split("//")[0].strip()
        if not line or line.endswith(":") or line.startswith(("Disassembly", "/")) or "file format" in line:
            continue
        yield line


def check(obj):
    ins = list(instructions(device_isa(obj)))
    mov = re.compile(r"^s_mov_b32 m0, s\d+$")
    bad, pairs = [], 0
    for i, t in enumerate(ins):
        if re.search(r"\bm0\b", t):
            ok = bool(mov.match(t)) and i + 2 < len(ins) and ins[i + 1].startswith("s_nop 0") and \
                re.match(r"^global_load_lds_dword(x4)?\b", ins[i + 2]) is not None
            pairs += ok
            if not ok:
                bad.append((i, t, ins[i + 1:i + 3]))
        if t.startswith(("global_load_lds", "buffer_load_lds")) or re.match(r"^(global|buffer)_load_\w+ .*\blds\b", t):
            if not (i >= 2 and mov.match(ins[i - 2]) and ins[i - 1].startswith("s_nop 0")):
                bad.append((i, t, ins[max(0, i - 2):i]))
    if bad:
        for i, t, ctx in bad[:20]:
            print(f"check_m0: {obj}: instruction {i}: `{t}` (context {ctx})", file=sys.stderr)
        raise SystemExit(f"check_m0: {obj}: {len(bad)} M0 use(s) outside the s_mov / s_nop / global_load_lds triple of "
                         f"tile128_dma.h - the inline-assembly LDS-DMA is no longer safe in this translation unit")
    if pairs == 0:
        raise SystemExit(f"check_m0: {obj}: no LDS-DMA found at all - wrong object?")
    print(f"check_m0: {os.path.basename(obj)}: {pairs} LDS-DMA instructions, M0 touched by nothing else")


if __name__ == "__main__":
    for o in sys.argv[1:]:
        check(o)
