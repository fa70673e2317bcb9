"""VALU-write -> asm-MFMA-read hazard scan of the register-resident-weights conv (conv_zreg_kernel.h).

The MFMAs of that kernel are inline asm with AGPR operands, invisible to hipcc's hazard recogniser: a VALU write of a
register that one of the next instructions reads as an MFMA operand needs wait states (s_nop) the compiler will not
insert.  `scan(lines)` walks an ISA listing (hipcc -save-temps .s text, or `llvm-objdump -d` output) and returns every
VALU write that an MFMA within the next three instructions reads with fewer than 2 wait states in between.

    python profiles/tools/hazard_scan.py [file.s]        # exit code 1 if an unpadded pair exists
    python profiles/tools/hazard_scan.py --lib [libdelivr_hip.so]   # every conv3_zreg_kernel in the built library

`library_report(path)` (used by tests/test_isa_gate_cpu.py) extracts the gfx950 code objects from the shared library's
.hip_fatbin section, disassembles the z-reg kernels and returns, per kernel, the hazard count, the MFMA count and the
resource usage from the code-object metadata (VGPRs, AGPRs, scratch, SGPR spills).
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def _regs(tok):
    out = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]", tok):
        out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
    for m in re.finditer(r"\b([va])(\d+)\b", tok):
        out.add((m.group(1), int(m.group(2))))
    return out


def scan(raw_lines):
    """-> (findings, n_mfma): findings = [(line_no_of_write, write_text, line_no_of_mfma, mfma_text, wait_states)]"""
    lines = []
    for i, l in enumerate(raw_lines):
        t = l.split("//")[0].strip()  # llvm-objdump appends '// address: encoding'
        if t and not t.startswith((";", ".")) and not t.endswith(":"):
            lines.append((i + 1, t))
    findings, n_mfma = [], 0
    for idx, (ln, t) in enumerate(lines):
        if not t.startswith("v_mfma"):
            continue
        n_mfma += 1
        args = t.split(None, 1)[1].split(",")
        reads = set()
        for x in args[1:]:
            reads |= _regs(x)
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
                w = _regs(pt.split(None, 1)[1].split(",")[0])
                if w & reads and states < 2:
                    findings.append((pl, pt, ln, t, states))
            states += 1
    return findings, n_mfma


def scan_readback(raw_lines, min_states=12):
    """The opposite hazard: a NON-MFMA instruction (v_accvgpr_read, a VALU operand, a store) that reads a register an asm MFMA
    wrote fewer than `min_states` wait states earlier (an 8-pass XDL result needs ~11 before anything but an MFMA may read it;
    every instruction in between counts as one state, s_nop N as N + 1).  -> findings [(line_mfma, mfma, line_reader, reader, states)]"""
    lines = []
    for i, l in enumerate(raw_lines):
        t = l.split("//")[0].strip()
        if t and not t.startswith((";", ".")) and not t.endswith(":"):
            lines.append((i + 1, t))
    findings = []
    for idx, (ln, t) in enumerate(lines):
        if not t.startswith("v_mfma"):
            continue
        dst = _regs(t.split(None, 1)[1].split(",")[0])
        states = 0
        for fwd in range(1, min_states + 1):
            if idx + fwd >= len(lines):
                break
            pl, pt = lines[idx + fwd]
            parts = pt.split(None, 1)
            op = parts[0]
            if op == "s_nop":
                states += int(parts[1]) + 1
            else:
                if not op.startswith("v_mfma") and len(parts) > 1:
                    ops = parts[1].split(",")
                    # sources: every operand but the first of VALU / accvgpr instructions, every operand of stores
                    srcs = ops if op.startswith(("ds_write", "buffer_store", "global_store", "flat_store")) else ops[1:]
                    reads = set()
                    for x in srcs:
                        reads |= _regs(x)
                    if reads & dst and states < min_states:
                        findings.append((ln, t, pl, pt, states))
                        break
                if op.startswith("v_mfma") and _regs(parts[1].split(",")[0]) & dst:
                    break  # rewritten by a later MFMA: that one is checked on its own
                states += 1
            if states >= min_states:
                break
    return findings


def code_objects(lib_path, workdir):
    """gfx950 code objects inside the library's .hip_fatbin (one clang offload bundle per translation unit) -> file paths"""
    data = open(lib_path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, pos = [], 0
    while True:
        pos = data.find(magic, pos)
        if pos < 0:
            break
        n = int.from_bytes(data[pos + 24:pos + 32], "little")
        p = pos + 32
        for _ in range(n):
            off = int.from_bytes(data[p:p + 8], "little")
            size = int.from_bytes(data[p + 8:p + 16], "little")
            tl = int.from_bytes(data[p + 16:p + 24], "little")
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size > 0:
                f = os.path.join(workdir, f"co_{len(out)}.elf")
                open(f, "wb").write(data[pos + off:pos + off + size])
                out.append(f)
        pos += len(magic)
    return out


def _kernel_metadata(elf):
    """amdhsa.kernels entries of a code object: {symbol: {vgpr_count, agpr_count, private_segment_fixed_size, sgpr_spill_count, ...}}"""
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", elf], capture_output=True, text=True).stdout
    meta, cur = {}, None
    for line in txt.split("\n"):
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip().strip("'")
        if k == "agpr_count":  # first key of a kernel entry (the keys of an entry are sorted)
            cur = {"agpr_count": int(v)}
        elif cur is not None and k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "sgpr_spill_count", "vgpr_spill_count",
                                       "group_segment_fixed_size"):
            cur[k] = int(v)
        elif cur is not None and k == "symbol":
            meta[v[:-3] if v.endswith(".kd") else v] = cur  # (vgpr_count follows: same dict)
    return meta


def scan_m0(raw_lines):
    """Uses of m0 in a kernel's listing.  The LDS-DMA asm of conv_deep.hip writes m0 (`s_mov_b32 m0, sN` + `buffer_load ... lds`);
    hipcc treats m0 as reserved - a clobber on it is not honoured (-Winline-asm) - so the kernel is only correct while NOTHING
    ELSE in it depends on m0: no compiler-generated reader (s_movrel / v_movrel / s_sendmsg with m0 / ds_gws / LDS-DMA builtin)
    between our writes.  -> {"writes": n, "dma_reads": n, "other": [instruction text of every other line that mentions m0]}"""
    writes = dma = 0
    other = []
    for l in raw_lines:
        t = l.split("//")[0].strip()
        if not t:
            continue
        toks = re.split(r"[\s,]+", t)
        if re.search(r"\bm0\b", t):
            if toks[0] == "s_mov_b32" and toks[1] == "m0":
                writes += 1
            else:
                other.append(t)
        elif toks[0].startswith("buffer_load") and toks[-1] == "lds":
            dma += 1  # (reads m0 implicitly)
        elif toks[0].startswith(("s_movrel", "v_movrel", "ds_gws", "s_sendmsg")):
            other.append(t)
    return {"writes": writes, "dma_reads": dma, "other": other}


def library_report(lib_path, name_filter=("conv3_zreg_kernel", "conv3_zwino_kernel")):
    if isinstance(name_filter, str):
        name_filter = (name_filter,)
    rep = {}
    with tempfile.TemporaryDirectory() as wd:
        for elf in code_objects(lib_path, wd):
            syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "--wide", elf], capture_output=True, text=True).stdout
            names = [l.split()[-1] for l in syms.split("\n") if any(nf in l for nf in name_filter) and " FUNC " in l]
            if not names:
                continue
            meta = _kernel_metadata(elf)
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", elf], capture_output=True, text=True).stdout
            # split the listing per function symbol
            cur, body = None, {}
            for line in dis.split("\n"):
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
                if m:
                    cur = m.group(1)
                    body[cur] = []
                elif cur is not None:
                    body[cur].append(line)
            for nm in names:
                findings, n_mfma = scan(body.get(nm, []))
                back = scan_readback(body.get(nm, []))
                rep[nm] = {"hazards": len(findings), "first": findings[:3], "mfma": n_mfma, "readback_hazards": len(back),
                           "first_readback": back[:3], "m0": scan_m0(body.get(nm, [])), **meta.get(nm, {})}
    return rep


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--lib":
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "delivr_cfos_amd", "lib", "libdelivr_hip.so")
        bad = 0
        for k, v in sorted(library_report(lib).items()):
            print(k, {x: y for x, y in v.items() if x != "first"})
            bad += v["hazards"]
        sys.exit(1 if bad else 0)
    path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tmp/k.s"
    found, n = scan(open(path).read().split("\n"))
    for pl, pt, ln, t, states in found[:20]:
        print(f"line {pl}: {pt}   ->   line {ln}: {t}   (wait states between: {states})")
    print("MFMAs:", n, " unpadded VALU-write -> MFMA-read pairs:", len(found))
    sys.exit(1 if found else 0)
