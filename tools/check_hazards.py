"""Static check of the hand-placed wait states in the embedding kernels.

The split of an fp32 value into two fp16 planes uses v_fma_mixlo_f16 / v_fma_mixhi_f16 inside inline asm, which
the compiler's hazard recognizer does not look into; on gfx950 an MFMA needs two wait states after a VALU write
of one of its A/B/C operands.  This script disassembles the gfx950 code object embedded in an object file of the
library and reports every v_mfma that reads a register written by one of those instructions (or by
v_cvt_pkrtz_f16_f32 / v_cvt_pk_f16_f32, which feed the other plane) fewer than two wait states earlier, for every kernel
instantiation in the file.  Exit code 0 = none found.

usage: python tools/check_hazards.py [nmpc_amd/csrc/build/embed.o ...]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
WRITERS = ("v_fma_mixlo_f16", "v_fma_mixhi_f16", "v_cvt_pkrtz_f16_f32", "v_cvt_pk_f16_f32")
NEEDED = 2


def regs(tok):
    """VGPR numbers named by an operand token such as v12 or v[4:7] (AGPRs and SGPRs: none)."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def disassemble(obj):
    tmp = tempfile.mkdtemp()
    try:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, capture_output=True, cwd=tmp)
        co = [f for f in os.listdir(tmp) if "gfx950" in f]
        if not co:
            raise RuntimeError("no gfx950 code object in " + obj)
        return subprocess.run([OBJDUMP, "-d", os.path.join(tmp, co[0])], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def check(text):
    """-> (kernels seen, writer instructions seen, mfma instructions seen, list of findings)"""
    findings, kernels, nw, nm = [], 0, 0, 0
    func = None
    pending = []            # (register set, wait states seen since the write, text of the writer)
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            func = m.group(1); pending = []
            kernels += 1
            continue
        body = line.split("//")[0].strip()
        if not body or func is None:
            continue
        parts = body.replace(",", " ").split()
        op, args = parts[0], parts[1:]
        if op.startswith("v_mfma"):
            nm += 1
            read = set().union(*[regs(a) for a in args[1:4]]) if len(args) >= 4 else set()
            for rset, waited, wtxt in pending:
                if rset & read and waited < NEEDED:
                    findings.append("%s: '%s' reads v%s written by '%s' %d wait state(s) earlier" %
                                    (func, body, sorted(rset & read), wtxt, waited))
        step = 1
        if op == "s_nop" and args:
            step = int(args[0], 0) + 1
        pending = [(r, w + step, t) for r, w, t in pending if w + step < NEEDED + 4]
        if op in WRITERS and args:
            nw += 1
            pending.append((regs(args[0]), 0, body))
    return kernels, nw, nm, findings


def main(paths):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = paths or [os.path.join(root, "nmpc_amd", "csrc", "build", n) for n in ("embed.o", "embed_panel.o")]
    bad = 0
    for pth in paths:
        kernels, nw, nm, findings = check(disassemble(pth))
        print("%s: %d functions, %d split instructions, %d MFMAs, %d hazards" % (pth, kernels, nw, nm, len(findings)))
        for f in findings[:20]:
            print("  " + f)
        bad += len(findings)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
