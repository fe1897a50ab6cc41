#!/bin/bash
# forward / backward blend stage time against the number of resident waves per SIMD, capped through dynamic LDS
for pad in 0 3700 5000 7000 10000 17000; do
  echo "== forward LDS pad $pad"; GGD_BLEND_LDS_PAD=$pad CULL_MODES=1 python scripts/cull_ab.py 2 2>&1 | grep '^{' | cut -c1-150
done
for pad in 0 2000 4500 7000; do
  echo "== backward LDS pad $pad"; GGD_BLEND_BWD_LDS_PAD=$pad CULL_MODES=1 python scripts/cull_ab.py 2 2>&1 | grep '^{' | cut -c1-150
done
