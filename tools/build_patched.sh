#!/bin/bash
# tools/build_patched.sh <name> <patch|-> [extra hipcc flags, e.g. -DMVP_EXP=3] -- compile a NON-product variant of the
# library from a patched COPY of the sources into build_variants/libmvp_<name>.so (the product tree is not touched).
# This is how the parked experiments of profiles/*.patch are rebuilt for an A/B on the GPU box
# (`python tools/bench_variant.py build_variants/libmvp_<name>.so ...`); "-" = no patch, flags only.
set -eu
cd "$(dirname "$0")/.."
NAME=$1; PATCH=$2; shift 2
TMP=$(mktemp -d)
mkdir -p "$TMP/ava-256_amd" build_variants
cp -r ava-256_amd/csrc "$TMP/ava-256_amd/csrc"
cp -r include "$TMP/include"
if [ "$PATCH" != "-" ]; then
  case "$PATCH" in /*) P="$PATCH";; *) P="$PWD/$PATCH";; esac; (cd "$TMP" && patch -p1 -s < "$P")
fi
SRCS=$(python3 - <<'PY'
import importlib.util, os
spec = importlib.util.spec_from_file_location("b", os.path.join("ava-256_amd", "build.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
print(" ".join(m.SOURCES)); print(" ".join(f for f in m.FLAGS))
PY
)
FILES=$(echo "$SRCS" | sed -n 1p); FLAGS=$(echo "$SRCS" | sed -n 2p)
hipcc $FLAGS "$@" -I "$TMP/include" -I "$TMP/ava-256_amd/csrc" $(for f in $FILES; do echo "$TMP/ava-256_amd/csrc/$f"; done) \
  -o "build_variants/libmvp_$NAME.so"
rm -rf "$TMP"
ls -la "build_variants/libmvp_$NAME.so"
