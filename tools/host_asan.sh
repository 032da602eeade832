#!/bin/bash
# Host-side AddressSanitizer pass (CPU container; GPU ASan is not available on the pool): the host-heavy objects -- ingest (gzip, index,
# ranged parse, FASTA scan / pack), the table formatter's host path, the switch table, the window classifier -- are rebuilt with
# -fsanitize=address on the HOST side only (-Xarch_host), linked with the other objects of csrc/ into /tmp/asan/libmural_hip_debug.so,
# and the CPU tests that drive them run on that library under the clang ASan runtime.  No report = exit status 0 of pytest.
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
CS=$REPO/mural_amd/csrc
OUT=/tmp/asan
RT=$(find /opt/rocm/lib/llvm -name "libclang_rt.asan-x86_64.so" | head -1)
mkdir -p $OUT
make -C $CS -j8 > /dev/null
for f in ingest tsv encode; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer \
    -ffp-contract=off -fno-honor-nans -c $CS/$f.hip -o $OUT/$f.o
done
# (clang refuses target_clones + visibility on one declaration; the sanitizer build drops the visibility)
sed 's/, visibility("hidden")//' $CS/host_classify.cpp > $OUT/hc.cpp
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fPIC -fsanitize=address -fno-omit-frame-pointer -c $OUT/hc.cpp -o $OUT/host_classify.o
OTHERS=$(cd $CS && ls *.o | grep -v -E "^(ingest|tsv|encode|host_classify)\.o$" | sed "s#^#$CS/#")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Xarch_host -fsanitize=address -shared-libsan $OTHERS $OUT/ingest.o $OUT/tsv.o $OUT/encode.o \
  $OUT/host_classify.o -lz -o $OUT/libmural_hip_debug.so
cd $REPO
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 MURAL_HIP_FLAVOR=debug python -c "
import mural_amd._lib as L
L.DEBUG_LIB_PATH = '$OUT/libmural_hip_debug.so'
import pytest, sys
sys.exit(pytest.main(['tests/test_ingest.py', 'tests/test_tsv.py', 'tests/test_host_logic.py', 'tests/test_dist_gloo.py', '-q', '-m', 'not gpu',
                      '-k', 'not world and not part_file and not ranked_and_whole and not aligned_shards_give and not aligned_shards_find',
                      '-p', 'no:cacheprovider']))
"
