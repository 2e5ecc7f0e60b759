#!/usr/bin/env bash
# Variant libraries for timing ablations of agg_fwd_stream_kernel: the default objects + local_attn_aggregate.hip compiled
# with -DGFLA_AGG_ABL=<bits> (see kAggAbl in local_attn_aggregate.hip).  Output: tools/ubench/abl/libgfla_agg_abl<bits>.so
set -euo pipefail
cd "$(dirname "$0")/../../global_flow_local_attention_amd/csrc"
make -j8 > /dev/null
OUT=../../tools/ubench/abl; mkdir -p $OUT
for N in "$@"; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DGFLA_AGG_ABL=$N -c local_attn_aggregate.hip -o /tmp/agg_abl_$N.o
  OBJS=$(ls build/*.o | grep -v local_attn_aggregate.o)
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/agg_abl_$N.o -o $OUT/libgfla_agg_abl$N.so
  echo built $OUT/libgfla_agg_abl$N.so
done
