#!/bin/bash
# compile hourglass.hip alone (the development loop of the hourglass kernels) and show errors / warnings
cd "$(dirname "$0")/../deepfly3d_amd/csrc" && mkdir -p _obj && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast "$@" -c hourglass.hip -o _obj/hourglass.hip.o 2>&1 | grep -E "error|warning" -A3 | head -60
