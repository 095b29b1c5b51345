#!/usr/bin/env python3
"""Instruction mix of the main loop of each kernel in a gfx950 assembly file (hipcc -S --cuda-device-only):

    python tools/isa_mix.py file.s [kernel-name-substring ...]

Finds the longest backward-branch loop of every kernel and counts its instructions by issue class; prices them with the
measured issue costs of MI355X_MICROARCH.md ("vector-instruction ISSUE cost": plain VALU 4 cycles per wave-instruction when one
wave issues alone, 2 cycles (SIMD-32) when several waves share the SIMD; transcendentals and packed-fp32 2x that; DPP adds
counted at the plain rate) so that a VALU issue floor can be stated from instruction counts instead of from a guess."""
import collections
import re
import sys

TRANS = ("v_rcp", "v_exp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")


def classify(line):
    op = line.split()[0]
    if op.startswith("v_"):
        if any(op.startswith(t) for t in TRANS):
            return "valu_trans"
        if "dpp" in line:
            return "valu_dpp"
        if op.startswith("v_pk_"):
            return "valu_packed"
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane", "v_mov_b32_dpp")):
            return "valu_lane"
        if op.startswith("v_mfma"):
            return "mfma"
        return "valu_plain"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    txt = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2:]
    starts = [(i, m.group(1)) for i, l in enumerate(txt) for m in [re.match(r"^(_Z\w+):", l)] if m]
    for i, name in starts:
        if want and not any(w in name for w in want):
            continue
        j = i
        while j < len(txt) and "s_endpgm" not in txt[j]:
            j += 1
        body = txt[i:j]
        labels = {m.group(1): k for k, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        loops = []
        for k, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < k:
                loops.append((labels[m.group(1)], k))
        if not loops:
            continue
        a, b = max(loops, key=lambda t: t[1] - t[0])
        cls = collections.Counter()
        for l in body[a:b]:
            l = l.strip()
            if not l or l.startswith((".", ";")) or l.endswith(":"):
                continue
            cls[classify(l)] += 1
        valu = sum(v for k, v in cls.items() if k.startswith("valu"))
        # a transcendental and a packed-fp32 instruction hold the issue port twice as long as a plain one (packed fp32 has the
        # FLOP rate of plain fp32 on gfx950: tools/ubench/valu_rate.hip)
        weighted = valu + cls["valu_trans"] + cls["valu_packed"]
        print("%s\n  main loop: %d instructions: %s\n  VALU wave-instructions per trip %d (issue-weighted %d): %.0f cycles at 2 cycles "
              "(several waves per SIMD), %.0f at 4 (one wave alone)" % (name, sum(cls.values()), dict(cls), valu, weighted,
                                                                       2.0 * weighted, 4.0 * weighted))


if __name__ == "__main__":
    main()
