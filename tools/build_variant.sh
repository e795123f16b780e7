#!/bin/bash
# Builds a variant of the library from the working tree with extra compiler flags into path-tracer_amd/lib/libptr_hip_NAME.so
# (for tools/ab3.sh):  bash tools/build_variant.sh NAME "-DPTR_FAST_PPL=4 -DPTR_FAST_WAVES=5"
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; EXTRA=${2:-}
D=/tmp/ptr_var_$NAME
rm -rf $D; mkdir -p $D/path-tracer_amd/lib $D/include
cp -r $R/path-tracer_amd/csrc $D/path-tracer_amd/csrc
cp $R/include/ptr.h $D/include/
rm -f $D/path-tracer_amd/csrc/*.o
make -C $D/path-tracer_amd/csrc -j8 EXTRA="$EXTRA" > $D/build.log 2>&1 || { tail -5 $D/build.log; exit 1; }
cp $D/path-tracer_amd/lib/libptr_hip.so $R/path-tracer_amd/lib/libptr_hip_$NAME.so
echo "built libptr_hip_$NAME.so ($EXTRA)"
