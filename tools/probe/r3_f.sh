cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_oamix.py -m gpu -x -q 2>&1 | tail -3
OADG_OAMIX_PLAN_C=0 python -m pytest tests/test_hip_oamix.py -m gpu -x -q 2>&1 | tail -2
bash tools/probe/ab_env.sh OADG_OAMIX_PLAN_C 0 1 3
