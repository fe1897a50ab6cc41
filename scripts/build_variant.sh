#!/bin/bash
# build_variant.sh <git-rev> <out.so>: the library as of <git-rev> (csrc + include), for A/B timing via GGD_LIB_PATH
set -e
REV=$1; OUT=$2
T=$(mktemp -d)
git -C "$(dirname "$0")/.." archive "$REV" gaussian_gan_decoder_amd/csrc include | tar -x -C "$T"
cd "$T/gaussian_gan_decoder_amd/csrc"
OBJS=""
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -munsafe-fp-atomics -fno-gpu-rdc -I"$T/include" -I. -c "$f" -o "${f%.hip}.o" &
  OBJS="$OBJS ${f%.hip}.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc $OBJS -o "$OUT"
rm -rf "$T"
echo built "$OUT" from "$REV"
