cd $GRAFT_REPO_ROOT
echo "== lockstep (default)"; timeout -k 10 200 python tools/time_two_view.py 2>&1 | tail -4 || exit 1
echo "== interleaved (CLC_ACR_LOCKSTEP=0)"; CLC_ACR_LOCKSTEP=0 timeout -k 10 200 python tools/time_two_view.py 2>&1 | tail -4 || exit 1
