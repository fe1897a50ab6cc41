cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_losses.py -m gpu -q 2>&1 | tail -12 | cut -c1-220
timeout 300 python scripts/profile_train.py --fused 2>&1 | tail -1
timeout 300 python scripts/profile_train.py --fused --torch-loss 2>&1 | tail -1
