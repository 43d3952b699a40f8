#!/bin/bash
# Builds a variant of the library with extra compiler flags for ONE source file, for kernel A/B runs on the GPU box:
#   scripts/build_variant.sh <name> <file.hip> -DFOO [-DBAR ...]   ->  lidog_amd/_C/variants/liblidog_<name>.so
# (the other objects are the plain build's).  Run with LIDOG_SO=$PWD/lidog_amd/_C/variants/liblidog_<name>.so.
# Experiment switches live in the sources only while an experiment is being measured; the shipped library is always the
# plain build of lidog_amd/build.py.
set -e
name=$1; file=$2; shift 2
cd "$(dirname "$0")/.."
python -c "from lidog_amd import build; build.build()" > /dev/null
out=lidog_amd/_C/variants; mkdir -p $out
obj=$out/${file%.hip}_$name.o
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function -ffp-contract=off "$@" -c lidog_amd/csrc/$file -o $obj
# the link list is the plain build's SOURCE list (lidog_amd/build.py), never a glob: a stray object in _C/ (an older
# variant, an experiment) must not be linked next to, or instead of, the plain one
objs=""
for src in $(python -c "from lidog_amd import build; print(' '.join(build.SOURCES))"); do
  f=lidog_amd/_C/${src%.hip}.o
  [ "$src" = "$file" ] && objs="$objs $obj" || objs="$objs $f"
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/liblidog_$name.so $objs -ldl
rm -f $obj
echo $out/liblidog_$name.so
