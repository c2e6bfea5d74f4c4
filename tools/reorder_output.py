#!/usr/bin/env python3
"""Sort a threaded driver's stdout by pair number (the job of the reference's scripts/reorderOutput.py:7-65).

    python tools/reorder_output.py <in> <out>

Header lines (before the first "<n> | <score>" block) and footer lines ("Elapsed time", "Cleaning up", statistics)
keep their places; result blocks ("<n> | <score>" + three lines, which may be empty) are emitted in pair order.
dpx_main prints in order already; this is for dpx_class_main / the reference's main.cpp, whose 20 pthreads race for stdout.
"""
import re
import sys

BLOCK = re.compile(r"^(\d+) \| (-?\d+)$")


def reorder(text: str) -> str:
    lines = text.split("\n")
    header, blocks, footer, i = [], {}, [], 0
    while i < len(lines):
        m = BLOCK.match(lines[i])
        if m:
            blocks[int(m.group(1))] = lines[i:i + 4]
            i += 4
        else:
            (footer if blocks else header).append(lines[i])
            i += 1
    out = header
    for k in sorted(blocks):
        out += blocks[k]
    return "\n".join(out + footer)


if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    with open(sys.argv[1], encoding="latin-1") as f:
        data = f.read()
    with open(sys.argv[2], "w", encoding="latin-1") as f:
        f.write(reorder(data))
