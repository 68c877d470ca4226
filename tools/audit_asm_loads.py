#!/usr/bin/env python3
"""Audit of the hidden (inline-asm) register loads in the device assembly of gemm.hip.

The compiler treats the destination of an asm load as written when the statement ends, so it is free to copy,
spill or reuse those registers before the data has landed.  This script walks the .s text in program order and
reports every instruction that touches the destination registers of an asm `global_load_*` between the load and
the hand-written `s_waitcnt vmcnt(N) ; hidden-load wait` that releases it (a counted wait releases the oldest pending load, vmcnt(0)
all of them), and checks every counted wait: N must not exceed the number of vector-memory instructions issued after
the load it releases, or the wait could return before the data has landed.  The load of a tile's mask bits ahead of
the phased kernel's K loop has no wait of its own (the loop's counted waits cover it, being younger): it is released by
the asm comment "tile mask bits arrived" after the loop, and nothing may touch its registers before that.

usage: audit_asm_loads.py gemm.s      (hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S gemm.hip)
"""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def main(path):
    kernel, in_asm, pending, bad, loads, vm_seq = None, False, [], 0, 0, 0
    markers_without_load = {}
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        if s.startswith("_Z") and ":" in s.split(";")[0]:
            kernel, pending = s.split(":")[0], []
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if in_asm and s.startswith("; tile mask bits arrived"):
            if pending:
                pending.pop(0)
            else:
                markers_without_load[kernel] = markers_without_load.get(kernel, 0) + 1
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        code = s.split(";")[0]
        if re.match(r"(global|buffer|flat|scratch)_(load|store|atomic)", code):
            vm_seq += 1
        if in_asm and code.startswith("global_load_"):
            dst = code.split(",")[0]
            # the address operand must not be the destination of an older pending load; its own destination may overlap it
            addr = regs_of(",".join(code.split(",")[1:]))
            for at, d, _ in pending:
                if addr & d:
                    bad += 1
                    print(f"{path}:{ln}: {kernel}: address uses v{sorted(addr & d)} of the load at line {at} before its wait: {s}")
            pending.append((ln, regs_of(dst), vm_seq))
            loads += 1
            continue
        elif in_asm and code.startswith("s_waitcnt vmcnt(") and "hidden-load wait" in s:
            n = int(code[len("s_waitcnt vmcnt("):].split(")")[0])
            if n == 0:
                pending = []
            elif pending:
                at, _, seq = pending.pop(0)
                if n > vm_seq - seq:
                    bad += 1
                    print(f"{path}:{ln}: {kernel}: vmcnt({n}) releases the load at line {at}, but only {vm_seq - seq} vector-memory instructions follow it")
            continue
        if code.startswith("s_endpgm"):
            pending = []
        used = regs_of(code)
        for at, dst, _ in pending:
            hit = used & dst
            if hit:
                bad += 1
                print(f"{path}:{ln}: {kernel}: touches v{sorted(hit)} of the load at line {at} before its wait: {s}")
    # the kernels that read a tile-ordered mask (epilogue 6) must pair every marker with a load
    for k, n in markers_without_load.items():
        if "ELi6EE" in k:
            bad += n
            print(f"{path}: {k}: {n} mask markers without a pending load")
    print(f"{loads} hidden loads, {bad} violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
