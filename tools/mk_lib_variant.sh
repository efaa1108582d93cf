#!/bin/bash
# a full build of the library under extra compiler switches -> flash_hash_join_amd/lib/ab/<name>.so (loaded with FJ_LIB_VARIANT=<name>):
#   tools/mk_lib_variant.sh <name> -DFJ_LAB [-DFJ_WIDE_BS=8 ...]
cd "$(dirname "$0")/../flash_hash_join_amd/csrc" || exit 1
name=$1; shift
mkdir -p ../lib/ab /tmp/fjv_$name
make -s -j4 OUT=../lib/ab/$name.so OBJS="$(for f in fj_partition fj_join fj_join_wide fj_bloom fj_many fj_pack fj_plan fj_joins fj_stream fj_bcast fj_hostentry fj_dist; do printf '/tmp/fjv_%s/%s.o ' $name $f; done)" EXTRA="$*" VARIANT_DIR=/tmp/fjv_$name variant && echo "built lib/ab/$name.so"
