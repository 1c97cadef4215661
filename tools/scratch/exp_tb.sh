cd $GRAFT_REPO_ROOT
for tb in 0 512 384 640; do
  for rep in 1 2; do
    if [ $tb = 0 ]; then unset CLC_K2NN_TARGET_BLOCKS; else export CLC_K2NN_TARGET_BLOCKS=$tb; fi
    timeout -k 10 200 python bench.py --headline-only --steps 200 --settle-steps 1500 > gpurun_out/exp_tb_${tb}_$rep.json 2> gpurun_out/exp_tb_${tb}_$rep.err || { echo fail $tb; tail -3 gpurun_out/exp_tb_${tb}_$rep.err; exit 1; }
    python3 -c "
import json,sys
d=json.loads(open('gpurun_out/exp_tb_${tb}_$rep.json').read().strip().splitlines()[-1])
print('target_blocks', '$tb', 'ms_per_step %.4f' % d['ms_per_step'], 'one_stream %.4f' % d['one_stream']['ms_per_step'], 'sweep_us %.1f' % d['stages']['k2nn_sweep_us'], 'overlapped sweep %.1f' % d['pipelined']['sweep_us_while_overlapped'])
"
  done
done
