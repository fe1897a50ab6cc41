#!/bin/bash
# build_variant.sh <git-rev | WORK> <out.so> [extra hipcc flags...]: the library as of <git-rev> (or the working tree), for A/B
# timing of kernel variants inside one gpurun call (select with GGD_LIB_PATH=<out.so>)
set -e
REV=$1; OUT=$2; shift 2
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
T=$(mktemp -d)
if [ "$REV" = WORK ]; then
  mkdir -p "$T/gaussian_gan_decoder_amd"; cp -r "$ROOT/gaussian_gan_decoder_amd/csrc" "$T/gaussian_gan_decoder_amd/"; cp -r "$ROOT/include" "$T/"
else
  git -C "$ROOT" archive "$REV" gaussian_gan_decoder_amd/csrc include | tar -x -C "$T"
fi
cd "$T/gaussian_gan_decoder_amd/csrc"
OBJS=""
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -munsafe-fp-atomics -fno-gpu-rdc -I"$T/include" -I. "$@" -c "$f" -o "${f%.hip}.o" 2>/dev/null &
  OBJS="$OBJS ${f%.hip}.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc $OBJS -o "$OUT"
rm -rf "$T"
echo built "$OUT" from "$REV" "$@"
