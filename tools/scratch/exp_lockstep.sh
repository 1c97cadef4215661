cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_two_view_batch.py tests/test_gpu_acransac.py -x -q 2>&1 | tail -4 || exit 1
echo "== lockstep (default)"; timeout -k 10 200 python tools/time_pose_batch2.py 2>&1 | tail -6 || exit 1
echo "== interleaved (CLC_ACR_LOCKSTEP=0)"; CLC_ACR_LOCKSTEP=0 timeout -k 10 200 python tools/time_pose_batch2.py 2>&1 | tail -6 || exit 1
