#!/bin/bash
# tools/build_variant.sh NAME "-DGIDDY_...=..."  -> variants/NAME.so  (development aid)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/variants
make -C $ROOT/libgiddy_amd/csrc -j8 -s BUILD=$ROOT/build/variant_$1 TARGET=$ROOT/variants/$1.so EXTRA="$2"
