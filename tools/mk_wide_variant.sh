#!/bin/bash
# a library variant that differs from the in-tree build in csrc/fj_join_wide.hip's compile-time switches only:
#   tools/mk_wide_variant.sh <name> -D<SWITCH>=<value> ...   ->  flash_hash_join_amd/lib/ab/<name>.so   (loaded with FJ_LIB_VARIANT=<name>)
cd "$(dirname "$0")/../flash_hash_join_amd/csrc" || exit 1
name=$1; shift
mkdir -p ../lib/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c fj_join_wide.hip -o /tmp/fj_join_wide_$name.o || exit 1
objs=$(ls ../lib/obj/*.o | grep -v fj_join_wide.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/ab/$name.so $objs /tmp/fj_join_wide_$name.o -ldl && echo "built lib/ab/$name.so"
