#!/bin/bash
# timing experiments on sort_place_stream (GULON_PLACE_DBG bit mask: 1 no stores, 2 no LDS scatter, 8 no slice loads)
for d in "$@"; do
  GULON_PLACE_DBG=$d scripts/upd_ab.sh pdbg_$d 1 | grep "sort_place" | tail -1 | sed "s/^/dbg=$d /"
done
