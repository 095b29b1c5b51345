#!/usr/bin/env python3
"""profiles/roundN_parity_passrates.txt from one `pytest tests -m gpu -s` log: every line the parity tests
print about the distance to north_star's PLAIN tolerance (1e-3 relative) -- pass-rates of gradient elements, the conditioning of
the fp32 problem (|hip - f64| beside |f32 oracle - f64|), imposed-decision counts, the free-running separation curves -- grouped
under the test that printed it.

    python tools/parity_report.py gpurun_out/<run>/pytest_s.log > profiles/round6_parity_passrates.txt
"""
import re
import sys

KEYS = ("within 1e-3", "conditioning", "gradient error", "gradients (leaf", "imposed", "disagree", "separation", "free-running",
        "pass-rate", "pass rate", "loss curve", "kink", "rel_l2", "relative L2", "host ms per step")


def main():
    lines = open(sys.argv[1], errors="replace").read().splitlines()
    print("# distance of the HIP path to the plain 1e-3 tolerance, as printed by the parity tests of one `pytest tests -m gpu -s` run")
    print("# source log: %s" % sys.argv[1])
    tail = [ln for ln in lines if re.search(r"\d+ passed", ln)]
    if tail:
        print("# suite: %s" % tail[-1].strip("= "))
    print()
    n = 0
    for ln in lines:
        body = ln.lstrip(".FsxE")
        if any(k in body for k in KEYS) and not body.startswith(("E ", ">", "tests/")):
            print(body.strip())
            print()
            n += 1
    print("# %d report lines" % n)


if __name__ == "__main__":
    main()
