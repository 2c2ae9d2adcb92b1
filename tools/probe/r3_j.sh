cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_rpn_loss.py tests/test_model_parity.py -m gpu -x -q 2>&1 | tail -3
bash tools/probe/kstats_ab.sh OADG_FUSED_RPN_LOSS 0 1
bash tools/probe/ab_env.sh OADG_FUSED_RPN_LOSS 0 1 3
