#!/bin/bash
# bench_bisect.sh: the train section of bench.py under different builds / options
cd "$(dirname "$0")/.."
ALL="--no-sweep --no-decode --no-inflight --no-extra-rooflines --no-cpu-baseline --steps 20"
run() { echo "== $*"; env "$@" timeout 600 python bench.py $ALL > /tmp/b.json 2> /tmp/b.err; echo "rc=$?"; grep -i "fault\|error\|Abort" /tmp/b.err | head -3; }
run GGD_MSD_SORT=0
run GGD_LIB_PATH=$PWD/variants_tmp/pp1.so
run GGD_LIB_PATH=$PWD/variants_tmp/pp2.so
run GGD_LIB_PATH=$PWD/variants_tmp/msd6.so
