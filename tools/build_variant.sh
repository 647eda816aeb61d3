#!/bin/bash
# usage: bash tools/build_variant.sh NAME "EXTRA_CXXFLAGS"  ->  ab/lib_NAME.so  (a copy of the sources built in /tmp; tools/ab.sh compares)
set -e
name=$1; flags=$2; root=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/gsr_var_$name && mkdir -p /tmp/gsr_var_$name/pkg /tmp/gsr_var_$name/include $root/ab
cp $root/include/*.h /tmp/gsr_var_$name/include/
mkdir -p /tmp/gsr_var_$name/pkg/csrc && cp $root/3dgs-native_amd/csrc/*.hip $root/3dgs-native_amd/csrc/*.h $root/3dgs-native_amd/csrc/Makefile /tmp/gsr_var_$name/pkg/csrc/
make -s -C /tmp/gsr_var_$name/pkg/csrc -j8 EXTRA="$flags"
cp /tmp/gsr_var_$name/pkg/libgsr_hip.so $root/ab/lib_$name.so
echo "built ab/lib_$name.so"
