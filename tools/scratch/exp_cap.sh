cd $GRAFT_REPO_ROOT
for cap in 128 16 8; do for ls in 1 0; do
echo "== cap $cap lockstep $ls"; CLC_ACR_BATCH_CAP=$cap CLC_ACR_LOCKSTEP=$ls timeout -k 10 200 python tools/time_two_view.py 8 2>&1 | tail -2 || exit 1
N=1000 OUTL=0.3 CLC_ACR_BATCH_CAP=$cap CLC_ACR_LOCKSTEP=$ls timeout -k 10 200 python tools/time_pose_batch2.py 2>&1 | grep -E "^GPU.* (1|4|8) solves" || exit 1
done; done
