set -u
TAG=r05
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG; cd /tmp && export TMPDIR=/tmp
python $R/scripts/profile_train.py --fused --standins > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/train -o p --output-format csv -- python $R/scripts/profile_train.py --fused --standins > $R/gpurun_out/$TAG/train.log 2>&1
python $R/scripts/profile_train.py --fused --fp32 --standins > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/train_fp32 -o p --output-format csv -- python $R/scripts/profile_train.py --fused --fp32 --standins > $R/gpurun_out/$TAG/train_fp32.log 2>&1
cd $R
rm -rf gpurun_out/$TAG/pmc gpurun_out/$TAG/pmc_shell
bash scripts/pmc_passes.sh $TAG/pmc scripts/fwd_only.py 1M_1024_cube 14 --backward > /dev/null
bash scripts/pmc_passes.sh $TAG/pmc_shell scripts/fwd_only.py 1M_1024_shell 14 --backward > /dev/null
python -m pytest tests/test_full_size_gpu.py -m gpu -q -x -s 2>&1 | grep -E "max\||passed|failed|dRGB" > gpurun_out/$TAG/full_size_errors.txt
tail -2 gpurun_out/$TAG/train_fp32.log | head -1; grep "ms/iter" gpurun_out/$TAG/train.log gpurun_out/$TAG/train_fp32.log
echo done
