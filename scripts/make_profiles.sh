#!/bin/bash
# Produce the round's rocprofv3 evidence on the GPU box (run through gpurun):  bash scripts/make_profiles.sh <tag>
# Outputs under gpurun_out/<tag>/ (copy the summaries into profiles/<tag>/ afterwards with profiles/collect.py):
#   bench/      rocprofv3 --kernel-trace --stats of the bench command (raster forward + backward, 100 steps, single stream:
#               --no-inflight, so that the kernels' average durations are the ones the bench line's roofline is priced on)
#   train/      the same for the fused-decoder train step (scripts/profile_train.py --fused)
#   pmc/        counter passes (scripts/pmc_passes.sh) of the raster forward + backward, 1 M / 1024^2 cube
#   pmc_shell/  ... of the shell scene
#   pmc_mlp/    ... of the fused decoder MLP, inference kernel, 1 M points (incl. the MFMA counters)
#   pmc_hl/     ... of the reference-precision decoder kernels (forward with z, backward, weight gradients), 2 M points
#   train_fp32/ kernel trace of the train step with the reference-precision fused decoder
#   hd/         kernel trace of scripts/hd_timing.py (1080p, 2048^2, 4K: every binning path that applies)
#   backward_blend_counters.txt, kernel_resources.txt, bench.json (plain `python bench.py`), full_size_errors.txt
set -u
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/bench -o p --output-format csv -- \
    python $R/bench.py --steps 100 --backward --no-train --no-decode --no-sweep --no-cpu-baseline --no-inflight --no-extra-rooflines \
    > $R/gpurun_out/$TAG/bench_under_rocprof.json 2> $R/gpurun_out/$TAG/bench.err
# (the perceptual stand-in's convolutions: an untraced run of the same script first, so that MIOpen's algorithm search -- its
# naive_conv* kernels were 93 % of round 4's table -- finds its results in the user database instead of running inside the
# traced process; MIOPEN_FIND_MODE=FAST would skip the search but pick slower kernels than the bench's step uses)
python $R/scripts/profile_train.py --fused --standins > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/train -o p --output-format csv -- \
    python $R/scripts/profile_train.py --fused --standins > $R/gpurun_out/$TAG/train.log 2>&1
python $R/scripts/profile_train.py --fused --fp32 --standins > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/train_fp32 -o p --output-format csv -- \
    python $R/scripts/profile_train.py --fused --fp32 --standins > $R/gpurun_out/$TAG/train_fp32.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/hd -o p --output-format csv -- \
    python $R/scripts/hd_timing.py > $R/gpurun_out/$TAG/hd_timing.txt 2>&1
cd $R
# (14 frames: the two-launch depth sort starts after 8 flat frames)
bash scripts/pmc_passes.sh $TAG/pmc scripts/fwd_only.py 1M_1024_cube 14 --backward > /dev/null
bash scripts/pmc_passes.sh $TAG/pmc_shell scripts/fwd_only.py 1M_1024_shell 14 --backward > /dev/null
bash scripts/pmc_passes.sh $TAG/pmc_mlp scripts/mlp_only.py 40 > /dev/null
bash scripts/pmc_passes.sh $TAG/pmc_hl scripts/hl_only.py 3 > /dev/null
bash scripts/frame_traces.sh gpurun_out/$TAG > /dev/null 2>&1
python scripts/bwd_blend_stats.py > gpurun_out/$TAG/backward_blend_counters.txt 2> /dev/null
python scripts/kernel_resources.py > gpurun_out/$TAG/kernel_resources.txt 2> /dev/null
python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench_plain.err
python -m pytest tests/test_full_size_gpu.py -m gpu -q -x -s 2>&1 | grep -E "max\||passed|failed|dRGB" > gpurun_out/$TAG/full_size_errors.txt
echo done
