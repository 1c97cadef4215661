cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_describe.py tests/test_gpu_detect.py -x -q 2>&1 | tail -n 3
for rep in 1 2 3; do
COLOC_HIP_LIB=tools/bin/c_noswap.so python3 tools/time_describe.py 2>/dev/null | grep -E "rect area-weighted|rebuild"
python3 tools/time_describe.py 2>/dev/null | grep -E "rect area-weighted|rebuild"
done
