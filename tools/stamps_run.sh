#!/bin/bash
# in-kernel cycle stamps of the Gauss-Newton solve launch (run on the GPU box): builds the two objects that carry the stamps with -DIFX_STAMPS into a
# SCRATCH library (/tmp/libifx_stamps.so, selected through IFX_LIB) and prints tools/ktimes.py's table; instancefusion_amd/libifx.so is never touched
set -e
cd "$(dirname "$0")/.."
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-value -Wno-unused-result"
( cd instancefusion_amd/csrc && /opt/rocm/bin/hipcc $F -DIFX_STAMPS -c ifx_track.hip -o /tmp/t_s.o &
  cd instancefusion_amd/csrc && /opt/rocm/bin/hipcc $F -DIFX_STAMPS -c ifx_api.hip -o /tmp/a_s.o & wait )
( cd instancefusion_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libifx_stamps.so /tmp/a_s.o /tmp/t_s.o $(ls *.o | grep -v "ifx_api.o\|ifx_track.o") )
IFX_LIB=/tmp/libifx_stamps.so python tools/ktimes.py 5000000 40 2>&1 | tail -8
