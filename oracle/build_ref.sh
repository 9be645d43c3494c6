#!/bin/bash
# Builds the UNMODIFIED reference `biokanga` executable (csiro-crop-informatics/biokanga v4.4.2)
# from the sources where they lie under /root/reference, with plain g++ (the reference's autotools
# build system is not run).  Outputs go ONLY to oracle/_ref/ (git-ignored; travels to the GPU box).
# This binary is TEST INFRASTRUCTURE: it pins the C restatement in oracle/ and serves as the timed
# CPU baseline (bench.py cpu_baseline.kind == "reference").  Nothing in the product path uses it.
#
# Notes
#  * libbiokanga/sqlite3.c is absent from the reference mount (.MISSING_LARGE_BLOBS); the image's
#    real /usr/lib/x86_64-linux-gnu/libsqlite3.so.0 is linked instead (header libbiokanga/sqlite3.h
#    is present in the reference).  No stand-in code is written.
#  * Known reference bug (biokanga/Aligner.cpp:4822 vs :4810, stack-local thread args outliving the
#    frame when read loading takes > 3 s) is NOT patched: keep reference runs to inputs that load
#    in < 3 s (<= ~1 M reads of 100 bp).
set -euo pipefail
REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
OBJ="$OUT/obj"
[ -d "$REF/libbiokanga" ] || { echo "reference not present at $REF - nothing to build"; exit 0; }
mkdir -p "$OBJ/lib" "$OBJ/bk" "$OBJ/pl"
CXXFLAGS="-O2 -w -fpermissive -std=gnu++11"
JOBS=${JOBS:-8}

list() {
  for f in "$REF"/libbiokanga/*.cpp; do
    b=$(basename "$f" .cpp)
    case "$b" in stdafx|MemAlloc|str_*|DSsort|FMIndex|VisData|conservlib) continue;; esac
    echo "$f $OBJ/lib/$b.o -"
  done
  for f in "$REF"/biokanga/*.cpp; do
    b=$(basename "$f" .cpp)
    case "$b" in stdafx) continue;; esac
    echo "$f $OBJ/bk/$b.o -"
  done
  for f in "$REF"/libBKPLPlot/*.cpp; do
    b=$(basename "$f" .cpp)
    case "$b" in BKPlots) continue;; esac
    echo "$f $OBJ/pl/$b.o -"
  done
}
export CXXFLAGS
list | xargs -P "$JOBS" -L 1 bash -c '
  src=$0; obj=$1; extra=""
  case "$src" in */plstdio.cpp) extra="-DO_BINARY=0 -D_O_SHORT_LIVED=0 -D_O_TEMPORARY=0";; esac
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ]; then
    g++ $CXXFLAGS $extra -c "$src" -o "$obj" || exit 255
  fi'
g++ -o "$OUT/biokanga" "$OBJ"/lib/*.o "$OBJ"/bk/*.o "$OBJ"/pl/*.o \
    /usr/lib/x86_64-linux-gnu/libsqlite3.so.0 -lz -lpthread -ldl -lrt
echo "built $OUT/biokanga"
