#!/bin/sh
# Mirrors the reference's own golden DATA files (test outputs and input decks; no source code)
# for the hot path into tests/golden/reference/.  Run in the build container, where
# /root/reference exists; the GPU box only sees the committed copies.
set -e
R=/root/reference/regression
D="$(dirname "$0")/reference"
mkdir -p "$D"
cp $R/discretization/HGRAD/mrhyde.gold                 $D/discretization_HGRAD.gold
for c in 2D_verification 3D_verification 2D_verification_highorder 2D_verification_mpi 2D_verification_transient 2D_mixed_bcs; do
  cp $R/thermal/$c/mrhyde.gold  $D/thermal_$c.gold
  cp $R/thermal/$c/input.yaml   $D/thermal_$c.input.yaml
done
for c in Mixed Mixed_3d; do
  cp $R/porous/$c/mrhyde.gold  $D/porous_$c.gold
  cp $R/porous/$c/input.yaml   $D/porous_$c.input.yaml
done
cp $R/navierstokes/channel/mrhyde.gold  $D/navierstokes_channel.gold
cp $R/navierstokes/channel/input.yaml   $D/navierstokes_channel.input.yaml
