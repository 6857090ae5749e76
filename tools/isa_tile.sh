#!/bin/bash
# dump the ISA of one gs_tile_kernel instantiation and list its waits / memory ops (design aid)
cd /root/repo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -S --cuda-device-only -o /tmp/pm.s pyamg_amd/csrc/pamg_matrix.hip 2>&1 | grep " error" 
K=${1:-_ZN4pamg14gs_tile_kernelIdLi9ELi2EEEvNS_8TileArgsIT_EE}
awk "/^$K:/,/s_endpgm/" /tmp/pm.s > /tmp/tile2.s
wc -l /tmp/tile2.s
grep -n "s_waitcnt vmcnt(0)" /tmp/tile2.s
grep -A40 "\.amdhsa_kernel $K" /tmp/pm.s | grep "next_free_vgpr\|next_free_sgpr"
