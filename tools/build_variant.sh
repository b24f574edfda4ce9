#!/bin/bash
# A/B builds: tools/build_variant.sh <name> "<extra flags>" <file.hip> [...]  ->  vmp-for-svae_amd/lib/libvmp_hip_<name>.so
# (the named sources are recompiled with the extra flags, every other object comes from the regular build/;
#  select the variant at run time with VMP_LIB_PATH)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/vmp-for-svae_amd/csrc
name=$1; flags=$2; shift 2
mkdir -p $C/build_$name
objs=""
for f in $C/*.hip; do
  b=$(basename $f .hip)
  if [[ " $* " == *" $b.hip "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-variable -Wno-unused-but-set-variable $flags -c $f -o $C/build_$name/$b.o &
    objs="$objs $C/build_$name/$b.o"
  else
    objs="$objs $C/build/$b.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/vmp-for-svae_amd/lib/libvmp_hip_$name.so $objs -ldl
echo built $R/vmp-for-svae_amd/lib/libvmp_hip_$name.so
