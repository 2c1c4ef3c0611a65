#!/bin/bash
# A/B builds of the library: scripts/mkvariant.sh <name> "<extra hipcc flags>" <translation unit> [...]
# -> graphminer_amd/variants/lib_<name>.so (the named units recompiled with the flags, the rest taken from build/); run with scripts/ab.py
set -e
cd "$(dirname "$0")/../graphminer_amd"
name=$1; flags=$2; shift 2
mkdir -p variants build/var_$name
objs=""
for o in build/gm_*.o; do
  b=$(basename $o .o)
  if [[ " $* " == *" $b "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../include $flags -c csrc/$b.hip -o build/var_$name/$b.o
    objs="$objs build/var_$name/$b.o"
  else
    objs="$objs $o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o variants/lib_$name.so -Wl,-rpath,/opt/rocm/lib
echo "variants/lib_$name.so"
