#!/bin/bash
# Diagnostic: k_front's phases one at a time — measurement builds that return after phase A / B / C (-DFX_FRONT_STOP=1/2/3,
# feature_extraction_amd.build.build_variant("stopK", ["-DFX_FRONT_STOP=K"])) against the product library, the kernel alone on
# the chip (stage `k_prep` of tools/stage_times.py), for one scan and for the headline batch.
for B in 1 1024; do
  for lib in libfx_hip_stop1.so libfx_hip_stop2.so libfx_hip_stop3.so libfx_hip.so; do
    python tools/stage_times.py $lib $B 20 2>/dev/null | tr '\n' ' ' | sed -e 's/k_bucket.*//' ; echo " (batch $B)"
  done
done
