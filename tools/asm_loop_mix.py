#!/usr/bin/env python3
"""Instruction mix of the largest loop of each kernel in a gfx950 assembly file (hipcc -S --cuda-device-only).
usage: asm_loop_mix.py file.s [substring of the kernel symbol ...]"""
import collections, re, sys


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_rsq", "v_sqrt")): return "transcendental"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return op
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")): return "_".join(op.split("_")[:2])
    return op


def main():
    text = open(sys.argv[1]).read()
    wanted = sys.argv[2:]
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if wanted and not any(w in name for w in wanted):
            continue
        lines = [l.strip() for l in body.split("\n")]
        label_at = {l.split(":")[0]: i for i, l in enumerate(lines) if l.startswith(".LBB")}
        loops = []
        for i, l in enumerate(lines):
            mm = re.match(r"s_c?branch\w* (\.LBB\w+)", l)
            if mm and mm.group(1) in label_at and label_at[mm.group(1)] < i:
                loops.append((label_at[mm.group(1)], i))
        if not loops:
            continue
        a, b = max(loops, key=lambda x: x[1] - x[0])
        ins = [l.split()[0] for l in lines[a:b] if l and not l.startswith((".", ";"))]
        mix = collections.Counter(classify(op) for op in ins)
        print(f"{name}: largest loop {len(ins)} instructions")
        print("  ", dict(mix.most_common()))
        valu = collections.Counter(op for op in ins if classify(op) in ("valu", "transcendental"))
        print("  ", valu.most_common(24))


if __name__ == "__main__":
    main()
