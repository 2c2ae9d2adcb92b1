cd $GRAFT_REPO_ROOT
python -m pytest tests/test_model_parity.py tests/test_inference_path.py -m gpu -x -q 2>&1 | tail -3
bash tools/probe/ab_env.sh OADG_FUSED_PROPOSALS 0 1 3
