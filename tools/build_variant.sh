#!/bin/bash
# A second build of the library for a measurement or an experiment: the named kernel files compiled with extra flags, every other object
# taken from the regular build (make -C biokanga_amd/csrc first).
#   tools/build_variant.sh hist "-DBK_CAND_HIST" bk_wave.hip      -> biokanga_amd/lib/libbiokanga_amd_hist.so
# Run anything against it with BK_LIB=<that path> (biokanga_amd/binding.py).
set -eu
name=$1; flags=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/biokanga_amd/csrc
obj=$root/build/obj
var=$root/build/obj_$name
mkdir -p "$var"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
objs=""
for o in bk_index bk_prep bk_search bk_extend bk_wave bk_heavy bk_rescue bk_snp bk_sam bk_sa_build bk_image bk_engine bk_tune bk_exchange bk_snp_host bk_stream bk_upload sfx_file; do
  f=""
  for s in "$@"; do case $s in $o.hip|$o.cpp) f=$s ;; esac; done
  if [ -n "$f" ]; then
    x=""; case $f in *.cpp) [ $o = sfx_file ] || x="-x hip" ;; esac
    $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result $flags $x -c "$src/$f" -o "$var/$o.o"
    objs="$objs $var/$o.o"
  else
    objs="$objs $obj/$o.o"
  fi
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$root/biokanga_amd/lib/libbiokanga_amd_$name.so" $objs
echo "$root/biokanga_amd/lib/libbiokanga_amd_$name.so"
