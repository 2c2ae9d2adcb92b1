cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_conv.py -m gpu -x -q 2>&1 | tail -3
bash tools/probe/ab_env.sh OADG_PREP_BANK 0 1 3
