#!/usr/bin/env python3
"""Where the vector-memory instructions and waits of one kernel stand relative to its MFMAs, from a gfx950 assembly listing:

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -S --offload-device-only -Iinclude deepfly3d_amd/csrc/hourglass.hip -o /tmp/hg.s
    python scripts/vmem_map.py /tmp/hg.s bottleneck_wino_f32_kernelILb0ELb0ELb0E

One line per run of equal instructions: (MFMAs issued before it, opcode) count first-operands.  This is how round 6 found the residual loads of the
Winograd tail at MFMA 508 of 512 (hipcc sinks a compiler-visible load to its first use) instead of in front of the K loop where the source has them."""
import sys

path, sub = sys.argv[1], sys.argv[2]
L = open(path).read().split("\n")
st = next(i for i, l in enumerate(L) if l.startswith("_Z") and ":" in l and sub in l.split(":")[0])
en = next(i for i in range(st, len(L)) if L[i].startswith(".Lfunc_end"))
WATCH = ("global_load_dwordx4", "global_load_dwordx2", "global_load_dword", "global_store_dwordx4", "global_store_dword", "s_barrier",
         "global_load_lds_dwordx4", "scratch_load_dword", "scratch_store_dword")
mf, out = 0, []
for l in L[st:en]:
    t = l.strip().split()
    if not t:
        continue
    if t[0].startswith("v_mfma"):
        mf += 1
    if t[0] in WATCH or (t[0] == "s_waitcnt" and "vmcnt" in l):
        out.append((mf, t[0], " ".join(t[1:])[:44]))
prev, cnt = None, 0
for o in out:
    key = (o[0], o[1])
    if prev and key == prev[0]:
        cnt += 1
    else:
        if prev:
            print(prev[0], cnt, prev[1])
        prev, cnt = (key, o[2]), 1
if prev:
    print(prev[0], cnt, prev[1])
print("MFMAs in the listing:", mf)
