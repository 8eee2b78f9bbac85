"""CPU build check: `LRN_MFMA_INPLACE` issues v_mfma_f64_16x16x4_f64 through inline asm, which LLVM's hazard
recogniser does not model.  Compile csrc/gemm_f64.hip to gfx950 ISA with the Makefile's flags and check, around every
ASMSTART MFMA:
  * no VALU instruction within the 2 instructions before it writes one of its source registers (VALU write -> MFMA
    read needs wait states the compiler would otherwise insert);
  * none of the next 18 instructions that is not itself an MFMA touches its destination registers (an MFMA result read
    or overwritten by a VALU / memory instruction before the 16-pass operation has retired; the one intended read of the
    accumulators, the epilogue, sits behind LRN_MFMA_DRAIN's 48 wait states).
A violation would show up as silently wrong Schur entries after a compiler upgrade or a code edit, not as a crash."""
import os
import re
import shutil
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(__file__), "..", "loraine.jl_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), k) for k in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def split_ops(line):
    body = line.split(";")[0].strip()
    if not body or body.endswith(":") or body.startswith("."):
        return None
    parts = body.split(None, 1)
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return parts[0], ops


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_inline_asm_mfma_has_no_unmodelled_hazards(tmp_path):
    out = tmp_path / "gemm.s"
    subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "--cuda-device-only", "-S",
                    os.path.join(CSRC, "gemm_f64.hip"), "-o", str(out)], check=True, stderr=subprocess.DEVNULL)
    lines = [l.rstrip("\n") for l in open(out)]
    instr = [(i, split_ops(l)) for i, l in enumerate(lines)]
    instr = [(i, x) for i, x in instr if x]                      # (line number, (mnemonic, operands))
    pos = {i: k for k, (i, _) in enumerate(instr)}
    n_asm = 0
    for ln, l in enumerate(lines):
        if "ASMSTART" not in l:
            continue
        nxt = next(i for i in range(ln + 1, len(lines)) if split_ops(lines[i]))
        mn, ops = split_ops(lines[nxt])
        if not mn.startswith("v_mfma"):
            continue                                             # (s_nop drains and the like)
        n_asm += 1
        k = pos[nxt]
        dst, srcs = regs(ops[0]), regs(ops[1]) | regs(ops[2])
        for _, (pm, pops) in instr[max(0, k - 2):k]:
            if pm.startswith("v_") and not pm.startswith("v_mfma") and pops and regs(pops[0]) & srcs:
                raise AssertionError(f"line {nxt + 1}: `{pm} {', '.join(pops)}` writes a source of the asm MFMA just before it")
        waited = 0                                               # wait states since the MFMA issued (s_nop N = N + 1)
        for _, (am, aops) in instr[k + 1:k + 40]:
            if waited >= 18:
                break
            if am in ("s_branch", "s_endpgm", "s_setpc_b64"):   # (the fall-through text is another path)
                break
            if am == "s_nop":
                waited += int(aops[0]) + 1
                continue
            waited += 1
            if am.startswith("v_mfma") or am.startswith("s_"):
                continue
            if aops and any(regs(o) & dst for o in aops):
                raise AssertionError(f"line {nxt + 1}: `{am} {', '.join(aops)}` touches the accumulator of an asm MFMA in flight")
    assert n_asm > 100              # the masked loops of the three direct-to-LDS kernels are there at all
