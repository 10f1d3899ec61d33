#!/bin/bash
# Measurement builds of the library in scratch copies of the package (git-ignored, but they travel to the GPU box):
#   bash tools/build_variant.sh <name> "<extra hipcc flags>"     ->  variants/<name>/megatts2_amd/lib/libmegatts2_hip.so
# The in-tree library stays the production one.  Used by tools/gpu_round.sh ablate / clock.
set -e
NAME=$1; FLAGS=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=$ROOT/variants/$NAME
rm -rf "$D" && mkdir -p "$D"
cp -r "$ROOT/megatts2_amd" "$ROOT/include" "$D/"
rm -rf "$D/megatts2_amd/lib" "$D/megatts2_amd/__pycache__"
mkdir -p "$D/tools" && cp "$ROOT"/tools/*.py "$D/tools/"
(cd "$D" && MT2_EXTRA_HIPCC_FLAGS="$FLAGS" python -m megatts2_amd.build > build.log 2>&1)
echo "$FLAGS" > "$D/FLAGS"
rm -f "$D"/megatts2_amd/lib/*.o
ls -la "$D/megatts2_amd/lib/"
