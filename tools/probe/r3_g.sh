cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_conv.py tests/test_hip_proposals.py -m gpu -x -q 2>&1 | tail -3
bash tools/probe/kstats.sh r3g > /dev/null 2>&1
python tools/probe/kstats_show.py r3g 200 | grep -E "steps|sel_|rpn_|prep_weights_multi"
