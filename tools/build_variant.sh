#!/bin/bash
# A/B builds of libnis_hip.so for on-box comparisons:  tools/build_variant.sh NAME "-DFLAG=.. -DFLAG2"  ->  nf_amd/lib/variants/NAME.so
# (select with NIS_LIB=$PWD/nf_amd/lib/variants/NAME.so).  Only UNIT (default rambo) is rebuilt with the flags; the other
# objects are reused.   tools/build_variant.sh NAME "FLAGS" [UNIT]
set -e
cd "$(dirname "$0")/.."
UNIT=${3:-rambo}
OBJ=${4:-$UNIT}        # object the variant replaces (flow_wm.hip builds flow_wm4 .. flow_wm64: pass e.g. "flow_wm flow_wm32" with -DNIS_WM=32)
mkdir -p nf_amd/lib/variants /tmp/nisvar
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function $2 -c nf_amd/csrc/$UNIT.hip -o /tmp/nisvar/$1.o
OBJS=$(ls nf_amd/csrc/build/*.o | grep -v /$OBJ.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o nf_amd/lib/variants/$1.so $OBJS /tmp/nisvar/$1.o
echo nf_amd/lib/variants/$1.so
