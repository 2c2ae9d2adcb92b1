cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_roi_nms.py tests/test_analytic_known_answers.py -m gpu -x -q 2>&1 | tail -3
for a in 0 1 8 48; do OADG_TILE_ABL=$a python tools/probe/roi_bwd_probe.py 2>&1 | grep tiles; done
