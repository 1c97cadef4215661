cd $GRAFT_REPO_ROOT
for L in 2 3 4 2 3; do
  timeout -k 10 200 python bench.py --headline-only --lanes $L --steps 300 --settle-steps 1500 > gpurun_out/exp_lanes_$L.json 2> gpurun_out/exp_lanes_$L.err || { echo fail $L; tail -3 gpurun_out/exp_lanes_$L.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/exp_lanes_$L.json').read().strip().splitlines()[-1])
print('lanes', '$L', 'ms_per_step %.4f' % d['ms_per_step'], 'one_stream %.4f' % d['one_stream']['ms_per_step'], 'overlapped sweep %.1f' % d['pipelined']['sweep_us_while_overlapped'], 'identical', d['pipelined']['lanes_identical_results'], d['one_stream']['identical_results'])
"
done
