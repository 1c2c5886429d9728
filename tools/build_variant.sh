#!/bin/bash
# A/B aid: builds a copy of csrc/ with extra compile flags into tools/ab/<name>/libsfmba_hip.so (git-ignored; travels with gpurun);
# select it at run time with SFMBA_LIB=tools/ab/<name>/libsfmba_hip.so.     tools/build_variant.sh <name> "<extra flags>"
set -e
NAME=$1; EXTRA=$2
REPO=$(cd "$(dirname "$0")/.." && pwd)
DST=$REPO/tools/ab/$NAME
mkdir -p $DST/csrc $DST/../../../include 2>/dev/null || true
rm -rf $DST && mkdir -p $DST/sfm/csrc $DST/include
cp $REPO/sfm-toy-library_amd/csrc/*.hip $REPO/sfm-toy-library_amd/csrc/*.h $REPO/sfm-toy-library_amd/csrc/Makefile $DST/sfm/csrc/
cp $REPO/include/*.h $DST/include/
make -C $DST/sfm/csrc -s -j8 FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -fvisibility=hidden -Wall -Wno-unused-function $EXTRA"
cp $DST/sfm/csrc/libsfmba_hip.so $DST/libsfmba_hip.so
rm -rf $DST/sfm $DST/include
ls -la $DST
