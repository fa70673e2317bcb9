"""Scan the ISA of a kernel (gpurun_out/tmp/k.s, written by regions.sh) for VALU writes of a register that one of the next
three instructions reads as an operand of an asm v_mfma WITHOUT an s_nop in between: the asm MFMAs of conv_zreg.hip are
invisible to hipcc's hazard recognizer.  Prints every finding with the line number; exit code 1 if an unpadded one exists."""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tmp/k.s"
raw = open(path).read().split("\n")
lines = [(i + 1, l.strip()) for i, l in enumerate(raw) if l.strip() and not l.strip().startswith((";", "."))]


def regs(tok):
    out = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]", tok):
        out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
    for m in re.finditer(r"\b([va])(\d+)\b", tok):
        out.add((m.group(1), int(m.group(2))))
    return out


bad = 0
for idx, (ln, t) in enumerate(lines):
    if not t.startswith("v_mfma"):
        continue
    args = t.split(None, 1)[1].split(",")
    reads = set()
    for x in args[1:]:
        reads |= regs(x)
    states = 0
    for back in range(1, 4):
        if idx - back < 0:
            break
        pl, pt = lines[idx - back]
        op = pt.split(None, 1)[0]
        if op == "s_nop":
            states += int(pt.split()[1]) + 1
            continue
        if op.startswith("v_") and not op.startswith("v_mfma") and len(pt.split(None, 1)) > 1:
            w = regs(pt.split(None, 1)[1].split(",")[0])
            if w & reads and states < 2:
                bad += 1
                if bad <= 20:
                    print(f"line {pl}: {pt}   ->   line {ln}: {t}   (wait states between: {states})")
        states += 1
print("unpadded VALU-write -> MFMA-read pairs:", bad)
sys.exit(1 if bad else 0)
