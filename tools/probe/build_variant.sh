#!/usr/bin/env bash
# build_variant.sh NAME SRC.hip "EXTRA FLAGS": libvotenet_hip.so with one source recompiled under extra -D flags -> tools/probe/lib/libvotenet_NAME.so
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"; ROOT="$HERE/../.."; C="$ROOT/votenet_amd/csrc"
name="$1"; src="$2"; flags="$3"
mkdir -p "$HERE/lib" /tmp/variant_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -I"$ROOT/include" -c "$C/$src" -o /tmp/variant_$name/v.o
objs=$(ls "$C"/obj/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/variant_$name/v.o -o "$HERE/lib/libvotenet_$name.so"
echo "built $HERE/lib/libvotenet_$name.so"
