#!/bin/bash
# GPU box: pass-2 probes of the batched depth->cloud (1: arithmetic without stores, 2: stores without the divisions)
for v in 0 1 2; do
  touch 3dvision_amd/csrc/depth.hip
  TDV_HIPCC_FLAGS="-DDE_PROBE=$v $EXTRA" python 3dvision_amd/build.py > /dev/null || exit 1
  echo "== DE_PROBE=$v $EXTRA"
  bash tools/prof_kernels.sh dp tools/studies/depth_probe.py | grep "depth_bits\|emit_bits"
done
