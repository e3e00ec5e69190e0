#!/bin/bash
# Builds the UNMODIFIED reference qvz from the sources where they lie under /root/reference
# into oracle/_ref/ (git-ignored, travels to the GPU box like any other built artefact).
# Test infrastructure only: nothing in the product path may execute or link these outputs.
#
#   oracle/_ref/qvz_ref            reference with its own Makefile flags (src/Makefile:10: -O3 -DLINUX),
#                                  WELL state seeded from time(0)             (qv_stream.c:77-78)
#   oracle/_ref/qvz_ref_fixedseed  same flags + -DDEBUG kept at -O3: WELL state = 32 x 0x55555555
#                                  (qv_stream.c:79-83) -> deterministic .q / -u outputs
#   oracle/_ref/libqvzref.so       every reference object except main.o, for the harness in
#                                  oracle/ref_harness.c that dumps per-phase intermediates
set -euo pipefail
REF=${QVZ_REFERENCE:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
if [ ! -d "$REF/src" ]; then
  echo "build_ref: $REF not present (GPU box?) - keeping prebuilt files in $OUT" >&2
  exit 0
fi
mkdir -p "$OUT/obj" "$OUT/objdbg" "$OUT/objpic"
SRCS="well codebook main util lines quantizer pmf distortion qv_stream qv_compressor arith os_stream cluster"
for f in $SRCS; do
  gcc -O3 -w -I"$REF/include" -DLINUX        -c "$REF/src/$f.c" -o "$OUT/obj/$f.o"
  gcc -O3 -w -I"$REF/include" -DLINUX -DDEBUG -c "$REF/src/$f.c" -o "$OUT/objdbg/$f.o"
  if [ "$f" != main ]; then
    gcc -O3 -w -fPIC -I"$REF/include" -DLINUX -DDEBUG -c "$REF/src/$f.c" -o "$OUT/objpic/$f.o"
  fi
done
gcc "$OUT"/obj/*.o    -o "$OUT/qvz_ref"           -lc -lm -lrt
gcc "$OUT"/objdbg/*.o -o "$OUT/qvz_ref_fixedseed" -lc -lm -lrt
gcc -shared "$OUT"/objpic/*.o -o "$OUT/libqvzref.so" -lm -lrt
echo "build_ref: built $OUT/qvz_ref $OUT/qvz_ref_fixedseed $OUT/libqvzref.so"
gcc -O2 -w -I"$REF/include" -DLINUX "$HERE/ref_harness.c" -o "$OUT/ref_harness" -L"$OUT" -lqvzref -Wl,-rpath,'$ORIGIN' -lm -lrt
echo "build_ref: built $OUT/ref_harness"
# qvz_dropin: the reference's UNMODIFIED main.c + util.c linked against this repo's libqvzcompat.so (the reference's
# entry-point names implemented on the MI355X, include/qvz_compat.h) instead of the reference's own objects.
# It is the drop-in proof used by tests/test_gpu_dropin.py; rpath is relative so the binary works on the GPU box.
if [ -f "$HERE/../qvz_amd/libqvzcompat.so" ]; then
  gcc -O3 -w -I"$REF/include" -DLINUX "$REF/src/main.c" "$REF/src/util.c" -o "$OUT/qvz_dropin" \
      -L"$HERE/../qvz_amd" -lqvzcompat -lqvzhip -Wl,-rpath,'$ORIGIN/../../qvz_amd' -Wl,-rpath-link,/opt/rocm/lib -lm -lrt
  echo "build_ref: built $OUT/qvz_dropin (reference main.c on libqvzcompat.so)"
fi
