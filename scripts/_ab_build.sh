#!/bin/bash
# A/B builds of one kernel file: scripts/_ab_build.sh NAME [file.hip]  ->  transformerupscaler_amd/csrc/build/ab_NAME.so
set -e
cd "$(dirname "$0")/../transformerupscaler_amd/csrc"
f=${2:-fused_attn.hip}
b=${f%.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wall -Wno-unused-function $EXTRA -c $f -o build/${b}_ab_$1.o
objs=$(ls build/*.o | grep -v "_ab_" | grep -v "build/${b}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab_$1.so $objs build/${b}_ab_$1.o
echo built build/ab_$1.so
