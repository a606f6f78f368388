#!/usr/bin/env python3
"""Instruction histogram of one kernel in a gfx950 assembly listing (hipcc -S --offload-device-only).

    python scripts/isa_hist.py LISTING.s SUBSTRING [--top N] [--range LO:HI]

SUBSTRING selects the kernel by (mangled) name; the body is straight-line for the unrolled ring kernels, so the
static count is the per-wave dynamic count up to exec-masked tails.  Prints opcode classes (MFMA / VALU / LDS /
VMEM / SALU / waits) and the top opcodes, plus the register and LDS figures of the kernel descriptor.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "MFMA"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("v_"):
        return "VALU"
    if op in ("s_waitcnt", "s_barrier", "s_nop", "s_sleep", "s_setprio"):
        return "WAIT/" + op
    if op.startswith("s_"):
        return "SALU"
    return "other"


def main():
    path, sub = sys.argv[1], sys.argv[2]
    top = 40
    if "--top" in sys.argv:
        top = int(sys.argv[sys.argv.index("--top") + 1])
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and sub in l.split(":")[0] and l.rstrip().split(";")[0].strip().endswith(":"):
            start = i
            break
    if start is None:
        sys.exit(f"no kernel matching {sub!r}")
    end = start
    while end < len(lines) and not lines[end].lstrip().startswith("s_endpgm"):
        end += 1
    print(lines[start].split(":")[0])
    ops = collections.Counter()
    classes = collections.Counter()
    for l in lines[start + 1:end]:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        ops[op] += 1
        classes[classify(op)] += 1
    for c, n in classes.most_common():
        print(f"  {c:16s} {n}")
    print("  -- top opcodes")
    for op, n in ops.most_common(top):
        print(f"  {op:36s} {n}")
    # descriptor facts
    for l in lines[end:end + 400]:
        m = re.search(r"; (NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|LDSByteSize|NumSgprs|codeLenInByte)\b.*", l)
        if m:
            print("  " + l.strip("; ").strip())
        if l.startswith("_Z") and l.rstrip().endswith(":"):
            break


if __name__ == "__main__":
    main()
