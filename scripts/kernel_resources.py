"""Registers / scratch / occupancy of the hourglass kernels from hipcc's -Rpass-analysis=kernel-resource-usage remarks:
    cd deepfly3d_amd/csrc && hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -Rpass-analysis=kernel-resource-usage --cuda-device-only -c hourglass.hip -o /tmp/hg.o 2> /tmp/res.txt
    python scripts/kernel_resources.py /tmp/res.txt [substring ...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pats = sys.argv[2:]
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split()[0]
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    if pats and not any(p in name for p in pats):
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    scratch, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
    print(f"{name[:100]:100s} VGPR {g('VGPRs'):>3s} AGPR {g('AGPRs'):>3s} scratch {scratch:>4s} occ {occ}")
