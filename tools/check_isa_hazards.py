#!/usr/bin/env python3
"""Build-time check of the gfx950 code in a shared library for an instruction form that returns wrong results on MI355X.

    v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 with op_sel[1] = 1  (the LOW half takes src1's HIGH register)

gives a wrong low half in lanes 48-63, sporadically, while MFMA wavefronts of ANOTHER kernel run on the same compute unit (a second
stream is enough; measured with tools/probe/src/pk_opsel_hazard.hip, profiles/r05_pk_opsel_hazard.txt: every other operand selection,
v_pk_mov_b32 and the plain packed form are exact).  The compiler emits the form by itself when it packs two scalar multiply-adds that
share a multiplier held in the odd register of a pair (SLP vectoriser) -- and for f32x2 code with the broadcast operand written second.

    python tools/check_isa_hazards.py [votenet_amd/lib/libvotenet_hip.so]      exit status 1 when the form is present
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
BAD = re.compile(r"\bv_pk_(fma|mul|add)_f32\b.*\bop_sel:\[[01],1")


def hazards(lib):
    """-> list of (kernel symbol, instruction text) for every hazardous instruction in the library's gfx950 code objects."""
    objdump = os.path.join(LLVM, "llvm-objdump")
    if not os.path.exists(objdump):
        raise RuntimeError("llvm-objdump not found under %s" % LLVM)
    found = []
    tmp = tempfile.mkdtemp(prefix="isa_check_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([objdump, "--offloading", "lib.so"], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        objs = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
        if not objs:
            raise RuntimeError("no gfx950 code object found in %s" % lib)
        for f in objs:
            dis = subprocess.run([objdump, "-d", "--mcpu=gfx950", f], cwd=tmp, check=True, capture_output=True, text=True).stdout
            sym = "?"
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    sym = m.group(1)
                elif BAD.search(line):
                    found.append((sym, line.split("//")[0].strip()))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return found


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "votenet_amd", "lib", "libvotenet_hip.so")
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):  # a box without the ROCm LLVM tools: the build goes on, tests/test_abi.py checks where they exist
        print("check_isa_hazards: %s/llvm-objdump not found, %s NOT checked" % (LLVM, lib))
        return 0
    found = hazards(lib)
    if not found:
        print("check_isa_hazards: %s is clean" % lib)
        return 0
    per = {}
    for sym, ins in found:
        per.setdefault(sym, []).append(ins)
    for sym, ins in per.items():
        name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip() or sym
        print("%4d x in %s\n         e.g. %s" % (len(ins), name[:160], ins[0]))
    print("check_isa_hazards: %d packed f32 instructions take the low half's operand from src1's high register (see the header)" % len(found))
    return 1


if __name__ == "__main__":
    sys.exit(main())
