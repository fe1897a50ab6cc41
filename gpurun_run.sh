cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -4
timeout 600 python scripts/profile_train.py 2>&1 | grep -v amdgpu | cut -c1-200 | tail -24
