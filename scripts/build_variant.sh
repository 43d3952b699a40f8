#!/bin/bash
# scripts/build_variant.sh <name> <source.hip> <extra hipcc flags...>: liblidog_amd_<name>.so with ONE source rebuilt
# under extra flags (kernel A/B runs: LIDOG_SO=lidog_amd/_C/liblidog_amd_<name>.so python bench.py)
set -e
name=$1; src=$2; shift 2
C=lidog_amd/_C
python -m lidog_amd.build > /dev/null
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off "$@" -c lidog_amd/csrc/$src -o $C/${src%.hip}_$name.o
objs=$(ls $C/*.o | grep -v "_[a-z0-9]*\.o$" | grep -v "/${src%.hip}.o"; true)
objs=""
for o in coords sconv sconv_mfma bn bev conv2d conv2d_sparse data losses optim comm trunk hostprep; do
  if [ "$o" = "${src%.hip}" ]; then objs="$objs $C/${o}_$name.o"; else objs="$objs $C/$o.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $C/liblidog_amd_$name.so $objs -ldl
echo $C/liblidog_amd_$name.so
