#!/bin/bash
# Build the library of a git revision (default HEAD) next to the working tree's, for a same-box A/B:
#   bash tools/build_head_variant.sh [rev]  ->  smartpy_amd/csrc/libsmart_amd_prev.so   (then tools/ab_variants.sh)
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
mkdir -p $TMP/smartpy_amd/csrc $TMP/include
for f in $(git -C $ROOT ls-tree --name-only $REV smartpy_amd/csrc/ | grep -E '\.(hip|h|cpp)$'); do git -C $ROOT show $REV:$f > $TMP/$f; done
git -C $ROOT show $REV:include/smart_amd.h > $TMP/include/smart_amd.h
cd $TMP/smartpy_amd/csrc
H=/opt/rocm/bin/hipcc; C="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc"
$H $C -ffp-contract=off -c smart_literal.hip -o l.o 2>/dev/null && $H $C -ffp-contract=fast-honor-pragmas -fno-honor-nans -c smart_fast.hip -o f.o 2>/dev/null && \
$H $C -c smart_capi.hip -o c.o 2>/dev/null && OBJS="l.o f.o c.o" && \
{ [ -f smart_hostio.cpp ] && $H $C -pthread -c smart_hostio.cpp -o h.o && OBJS="$OBJS h.o"; true; } && \
$H -shared -fPIC -pthread --offload-arch=gfx950 -o $ROOT/smartpy_amd/csrc/libsmart_amd_prev.so $OBJS && ls -la $ROOT/smartpy_amd/csrc/libsmart_amd_prev.so
rm -rf $TMP
