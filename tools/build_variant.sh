#!/bin/bash
# Builds a VARIANT of the library beside the default one, for same-box A/B measurements:
#   tools/build_variant.sh NAME [XFLAGS...]   ->  deconv3d_amd/csrc/var_NAME/libdeconv3d_hip.so
# (use with DECONV3D_HIP_LIB=<that path>; XFLAGS are extra compiler flags, e.g. -DD3D_OLD_TAIL)
set -e
name=$1; shift
here=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p "$tmp/csrc" "$tmp/include" "$here/deconv3d_amd/csrc/var_$name"
cp "$here"/deconv3d_amd/csrc/*.hip "$here"/deconv3d_amd/csrc/*.h "$here"/deconv3d_amd/csrc/Makefile "$tmp/csrc/"
cp "$here/include/deconv3d_hip.h" "$tmp/include/"
sed -i 's#../../include/deconv3d_hip.h#../include/deconv3d_hip.h#' "$tmp/csrc/Makefile" "$tmp/csrc/d3d_ctx.h"
make -C "$tmp/csrc" -j3 XFLAGS="$*" >/dev/null
cp "$tmp/csrc/libdeconv3d_hip.so" "$here/deconv3d_amd/csrc/var_$name/"
rm -rf "$tmp"
echo "built $here/deconv3d_amd/csrc/var_$name/libdeconv3d_hip.so"
