#!/bin/bash
# in-kernel cycle stamps of the Gauss-Newton solve launch (run on the GPU box): rebuilds the two objects that carry the stamps with -DIFX_STAMPS into a
# scratch copy of the library, prints tools/ktimes.py's table, then restores the regular build
set -e
cd "$(dirname "$0")/.."
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-value -Wno-unused-result"
cp instancefusion_amd/libifx.so /tmp/libifx.keep
( cd instancefusion_amd/csrc && /opt/rocm/bin/hipcc $F -DIFX_STAMPS -c ifx_track.hip -o /tmp/t_s.o & 
  cd instancefusion_amd/csrc && /opt/rocm/bin/hipcc $F -DIFX_STAMPS -c ifx_api.hip -o /tmp/a_s.o & wait )
( cd instancefusion_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libifx.so /tmp/a_s.o /tmp/t_s.o ifx_map.o ifx_instance.o ifx_slic.o ifx_knn.o )
python tools/ktimes.py 5000000 40 2>&1 | tail -8
cp /tmp/libifx.keep instancefusion_amd/libifx.so
