"""tools/asm_async_lint.py -- static check of the inline-asm loads whose results arrive asynchronously.

csrc/gpx_gemm.hip issues its LDS fragment reads as `asm volatile("ds_read_b128 %0, ...": "=v"(dst))`; csrc/gpx_panel.hip loads
published blocks with `global_load_dwordx4 ... sc1` the same way.  For the compiler such a destination is written when the
statement has executed; for the hardware, when the data returns.  Nothing may read, copy or overwrite a destination register
between the load and the asm wait that covers it (round 6: an accumulator copy placed before the wait after the k-loop landed in
a register of the last fragment read -- one wrong fp32 fit in a thousand).  The sources tie the registers to their waits
(GPX_FRAG_WAIT, pub_wait); this tool reads the COMPILED code and checks that the result is what was meant:

    for every asm load (between ;;#ASMSTART / ;;#ASMEND) with destination registers D, walking forward along the fall-through
    path and every branch target until an asm `s_waitcnt` that drains the load's counter (lgkmcnt(0) / vmcnt(0)):
    no instruction may mention a register of D, except another asm load that overwrites it.

    python tools/asm_async_lint.py [file.hip ...]      (default: gpx_gemm.hip gpx_panel.hip; compiles with hipcc -S, ~1 min each)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gaussian_processes_amd", "csrc")

REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def parse(path):
    """-> {kernel: [(text, in_asm)]} with labels kept as instructions 'LABEL name'."""
    kernels, cur, in_asm = {}, None, False
    for line in open(path):
        t = line.strip()
        if not t:
            continue
        m = re.match(r"^(_Z\w+):", t)
        if m and "@function" not in t:
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        if t.startswith(".Lfunc_end"):
            cur = None
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if t.startswith(";") or t.startswith(".") and not t.startswith(".LBB"):
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            cur.append(("LABEL " + m.group(1), False))
            continue
        cur.append((t.split(";")[0].strip(), in_asm))
    return kernels


def lint_kernel(name, ins):
    labels = {t.split()[1]: i for i, (t, _) in enumerate(ins) if t.startswith("LABEL ")}
    problems, loads = [], 0
    for i, (t, asm) in enumerate(ins):
        if not asm:
            continue
        op = t.split()[0]
        if op.startswith("ds_read"):
            counter = "lgkmcnt(0)"
        elif op.startswith("global_load") and "lds" not in op:
            counter = "vmcnt(0)"
        else:
            continue
        dest = regs_of(t.split(",")[0])
        loads += 1
        # walk every path from i + 1 until the covering asm wait
        seen, stack = set(), [i + 1]
        while stack:
            k = stack.pop()
            while k < len(ins) and k not in seen:
                seen.add(k)
                tt, a = ins[k]
                o = tt.split()[0]
                if a and o == "s_waitcnt" and counter in tt:
                    break
                if o == "s_endpgm":
                    problems.append((name, i, t, k, "the program ends before a wait covers the load"))
                    break
                if o == "LABEL":
                    k += 1
                    continue
                touched = regs_of(tt) & dest
                if touched:
                    same_kind_load = a and (o.startswith("ds_read") or o.startswith("global_load")) and \
                        not (regs_of(",".join(tt.split(",")[1:])) & dest)
                    if not same_kind_load:
                        problems.append((name, i, t, k, tt))
                        break
                if o.startswith("s_cbranch") or o == "s_branch":
                    tgt = tt.split()[-1]
                    if tgt in labels:
                        stack.append(labels[tgt])
                    if o == "s_branch":
                        break
                k += 1
    return loads, problems


def main(files):
    total, bad = 0, 0
    for f in files:
        src = f if os.path.isabs(f) else os.path.join(CSRC, f)
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                                   "-o", out, src], stderr=subprocess.DEVNULL)
            kernels = parse(out)
        for name, ins in kernels.items():
            loads, problems = lint_kernel(name, ins)
            total += loads
            if loads:
                print("%s %s: %d asm loads, %d problems" % (os.path.basename(src), name[:70], loads, len(problems)))
            for p in problems[:6]:
                print("    load #%d  %s\n      -> before its wait, instruction #%d: %s" % (p[1], p[2], p[3], p[4]))
            bad += len(problems)
    print("asm loads checked: %d, problems: %d" % (total, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or ["gpx_gemm.hip", "gpx_panel.hip"]))
