#!/bin/bash
# One library BUILD VARIANT for profiles/ab_libs.sh: recompiles the named translation unit(s) with extra -D flags and links them with the
# tree's other objects into gpurun_ab/lib_<name>.so (git-ignored; travels to the GPU box).
#   profiles/build_variant.sh <name> "<-D flags>" [unit ...]     (default unit: zh_approx)
set -e
name=$1; flags=$2; shift 2
units=${@:-zh_approx}
cd "$(dirname "$0")/../zebra_amd/csrc"
make -s -j8 >/dev/null
mkdir -p ../../gpurun_ab /tmp/zh_variant_$name
objs=""
for o in zh_search zh_approx zh_order zh_score zh_build zh_api zh_shard zh_refformat; do
  if [[ " $units " == *" $o "* ]]; then
    src=$o.hip; [ -f $src ] || src=$o.cpp
    /opt/rocm/bin/hipcc $flags -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -c $src -o /tmp/zh_variant_$name/$o.o
    objs="$objs /tmp/zh_variant_$name/$o.o"
  else
    objs="$objs $o.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o ../../gpurun_ab/lib_$name.so $objs -L/opt/rocm/lib -lrccl
echo "built gpurun_ab/lib_$name.so ($flags; $units)"
