#!/bin/bash
# usage: tools/build_variant.sh <name> [-DFLAG ...]  -> tools/_build/libnerfsig_<name>.so (same sources and per-file flags as the product build, extra defines); select it with NERFSIG_LIB
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python -c "import sys; from nerf_signature_amd import build as b; print(b.build_variant(sys.argv[1], sys.argv[2:]))" "$name" "$@"
