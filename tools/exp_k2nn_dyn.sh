cd $GRAFT_REPO_ROOT
for rep in 1 2; do
CLC_K2NN_DYN=0 COLOC_HIP_LIB=tools/bin/k_dyn4.so python3 tools/time_match_one.py 2>/dev/null
for v in k_dyn3 k_dyn4; do COLOC_HIP_LIB=tools/bin/$v.so python3 tools/time_match_one.py 2>/dev/null; done
done
