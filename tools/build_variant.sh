#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG=.. ..."  ->  versecrafter_amd/libvcengine_NAME.so  (A/B builds; select with VC_ENGINE_LIB)
set -e
cd "$(dirname "$0")/../versecrafter_amd/csrc"
name=$1; shift
tmp=$(mktemp -d)
for f in $(sed -n 's/^SRCS *:\?= *//p' Makefile | sed 's/\.hip//g'); do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c $f.hip -o $tmp/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvcengine_$name.so $tmp/*.o -ldl
rm -rf $tmp
echo built versecrafter_amd/libvcengine_$name.so
