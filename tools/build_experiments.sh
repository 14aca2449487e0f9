#!/bin/bash
# Builds the EXPERIMENTS=1 variant of the library (phase stamps, k_mh_flow / k_mh_pair and the
# other measured-but-not-faster kernels) beside the default one:
#   deconv3d_amd/csrc/exp/libdeconv3d_hip.so     (use with DECONV3D_HIP_LIB=<that path>)
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p "$tmp/csrc" "$tmp/include" "$here/deconv3d_amd/csrc/exp"
cp "$here"/deconv3d_amd/csrc/*.hip "$here"/deconv3d_amd/csrc/*.h "$here"/deconv3d_amd/csrc/Makefile "$tmp/csrc/"
cp "$here/include/deconv3d_hip.h" "$tmp/include/"
sed -i 's#../../include/deconv3d_hip.h#../include/deconv3d_hip.h#' "$tmp/csrc/Makefile" "$tmp/csrc/d3d_ctx.h"
make -C "$tmp/csrc" -j3 EXPERIMENTS=1 XFLAGS="$XFLAGS" >/dev/null
cp "$tmp/csrc/libdeconv3d_hip.so" "$here/deconv3d_amd/csrc/exp/"
rm -rf "$tmp"
echo "built $here/deconv3d_amd/csrc/exp/libdeconv3d_hip.so"
