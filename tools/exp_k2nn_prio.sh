cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in k_base k_prio1 k_prio3 k_prio4 k_prio3p2; do COLOC_HIP_LIB=tools/bin/$v.so python3 tools/time_match_one.py 2>/dev/null; done
done
