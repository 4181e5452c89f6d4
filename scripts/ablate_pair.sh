#!/bin/bash
# Compile-time ablation builds of csrc/pair_block.hip (bits: 1 no weight DMA after the prologue, 2 no fragment reads, 4 no MFMAs,
# 8 no GELU, 16 no hook row I/O) -> prediff_amd/libprediff_hip_ab<N>.so (untracked); run on the GPU box:  [PD_BENCH_UNITS=512] bash scripts/ablate_pair.sh run [B]
cd "$(dirname "$0")/../prediff_amd/csrc" || exit 1
VARIANTS=${VARIANTS:-"0 1 2 4 8 3 6 12 15"}
if [ "$1" = "build" ]; then
  for v in $VARIANTS; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -DPD_PAIR_ABLATE=$v -c pair_block.hip -o /tmp/pair_ab$v.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o igemm256.o norm.o attention.o elementwise.o skill.o ffn.o attn_block.o conv2d_gn.o /tmp/pair_ab$v.o -o ../libprediff_hip_ab$v.so
  done
else
  cd ../..
  for v in $VARIANTS; do
    PD_LIB_PATH=$PWD/prediff_amd/libprediff_hip_ab$v.so timeout 200 python scripts/bench_pair.py ${2:-32} ablate 2>&1 | tail -1
  done
fi
