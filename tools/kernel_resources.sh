#!/bin/bash
# Register / scratch / LDS use of every kernel in a built object:  tools/kernel_resources.sh nf_amd/csrc/build/rambo.o [name-filter]
# (reads the code object's metadata: no recompile).  Occupancy by registers: waves/SIMD = min(8, 512 / roundup8(vgpr_count)).
set -e
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$1"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co | awk -v f="${2:-.}" '
  /\.group_segment_fixed_size:/ {lds=$2} /\.name:/ {name=$2} /\.private_segment_fixed_size:/ {scr=$2} /\.sgpr_count:/ {sg=$2}
  /\.vgpr_count:/ {vg=$2} /\.vgpr_spill_count:/ {sp=$2; if (name ~ f) printf "%-90s vgpr %3d sgpr %3d spill %3d scratch %4d lds %6d\n", name, vg, sg, sp, scr, lds}'
rm -rf $T
