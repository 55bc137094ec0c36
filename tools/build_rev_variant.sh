#!/bin/bash
# Build the library of a git revision next to the working tree's, for a same-box A/B (the revision must speak the
# working tree's ABI):   bash tools/build_rev_variant.sh <rev> <name>  ->  tools/variants/libsmart_amd_<name>.so
# then on the GPU box:   bash tools/ab_variants.sh tools/variants/libsmart_amd_<name>.so
REV=${1:-HEAD}; NAME=${2:-prev}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C $ROOT archive $REV smartpy_amd/csrc smartpy_amd/build.py include | tar -x -C $TMP
mkdir -p $ROOT/tools/variants
python3 - <<PY
import importlib.util, shutil
spec = importlib.util.spec_from_file_location('rev_build', '$TMP/smartpy_amd/build.py')
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
out = b.build(force=True, lib_path=b.LIB)
shutil.copy(out, '$ROOT/tools/variants/libsmart_amd_$NAME.so')
print('$ROOT/tools/variants/libsmart_amd_$NAME.so')
PY
rm -rf $TMP
