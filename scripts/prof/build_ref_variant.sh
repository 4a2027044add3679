#!/bin/bash
# Builds the library of another commit next to the in-tree one, for same-box A/B runs (ab_*.sh select it with
# VGGSFM_AMD_LIB):   scripts/prof/build_ref_variant.sh [git-ref, default HEAD~1]   ->  vggsfm_amd/_variants/lib_head.so
# (run in the build container; the .so travels to the GPU box with the snapshot, the directory is not tracked)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
REF=${1:-HEAD~1}
TMP=$(mktemp -d)
mkdir -p $TMP/vggsfm_amd $TMP/include $ROOT/vggsfm_amd/_variants
git -C $ROOT archive $REF vggsfm_amd/csrc include | tar -x -C $TMP
make -C $TMP/vggsfm_amd/csrc -j4 >/dev/null
cp $TMP/vggsfm_amd/libvggsfm_amd.so $ROOT/vggsfm_amd/_variants/lib_head.so
rm -rf $TMP
echo "built $REF -> vggsfm_amd/_variants/lib_head.so"
