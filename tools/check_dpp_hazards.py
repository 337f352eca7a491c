"""Build-time check of the DPP read-after-write hazard in the hand-written DPP FMAs (kernels.hip: fmac_row_bcast).

On gfx90a+ a VGPR written by a VALU instruction must not be read through DPP by one of the next two instructions.
The compiler's hazard recogniser pads real DPP instructions with s_nop but cannot see inside inline asm, so this
script compiles kernels.hip to assembly and verifies that no `*_dpp` instruction reads (src0) a register that a VALU
instruction wrote fewer than two wait states earlier (s_nop N counts N + 1).  Exit code 1 and the offending pairs on stdout if it finds one.
Usage: python tools/check_dpp_hazards.py [path/to/kernels.hip]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(asm_text):
    lines = [l.split(";")[0].strip() for l in asm_text.splitlines()]
    lines = [l for l in lines if l and not l.startswith(".") and not l.endswith(":")]
    ndpp, bad = 0, []
    for i, l in enumerate(lines):
        if "_dpp" not in l.split(None, 1)[0]:
            continue
        ndpp += 1
        ops = [t.strip() for t in l.split(None, 1)[1].split(",")]
        src0 = regs(ops[1].split()[0])
        ws, j = 0, i - 1          # wait states between the candidate writer and the DPP instruction (s_nop N = N + 1)
        while j >= 0 and ws < 2:
            p = lines[j]
            if p.startswith("v_"):
                dst = regs(p.split(None, 1)[1].split(",")[0].strip().split()[0])
                if dst & src0:
                    bad.append((p, l))
                    break
            m = re.match(r"s_nop\s+(\d+)", p)
            ws += int(m.group(1)) + 1 if m else 1
            j -= 1
    return ndpp, bad


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "commander_amd", "csrc", "kernels.hip")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", out],
                       check=True, cwd=td, stderr=subprocess.DEVNULL)
        ndpp, bad = scan(open(out).read())
    for p, l in bad:
        print("HAZARD:", p, "->", l)
    print("%d DPP instructions, %d hazards" % (ndpp, len(bad)))
    return 1 if bad or ndpp == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
