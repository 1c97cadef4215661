cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in "$@"; do
  OUT=gpurun_out/prof_det_$v; rm -rf $OUT; mkdir -p $OUT
  COLOC_HIP_LIB=tools/bin/$v.so timeout -k 5 60 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 tools/time_detect_one.py > $OUT/stdout.log 2>&1
  echo "== $v"; tail -n 1 $OUT/stdout.log
  find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} grep -E "detect_" {} | cut -d, -f1-4,6,7 | sed 's/clc::detect_//; s/(clc::DetectArgs[^"]*"/"/'
  find $OUT -type f ! -name stdout.log -delete
done
