#!/bin/bash
# Diagnostic build of the library with the fill kernel's wait counters (never shipped): writes
# pagan2-msa_amd/libpagan_dp_stats.so beside the product library; select it with PAGAN_DP_LIB=<that path>.
set -e
cd "$(dirname "$0")/../pagan2-msa_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -pthread -DPG_PIPE_STATS -DPG_TILE_STATS ${PG_STATS_EXTRA} -Wno-unused-result \
  -o ../libpagan_dp_stats.so dp_abi.hip dp_kernels.hip dp_pipe.hip dp_tiles.hip dp_fb.hip dp_anchors.hip dp_parent.hip host_model.cpp host_graph.cpp host_anchors.cpp host_tree.cpp host_pileup.cpp
