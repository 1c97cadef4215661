"""Soak: the virtual-world reassembly test of tests/test_gpu_multicam.py (8 ranks x 1200 keypoints) N times in one process.
usage: python tests/soak/multicam_virtual_world_loop.py [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import oracle_lib
import test_gpu_multicam as t

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
orc = oracle_lib.Oracle()
for it in range(n):
    t.test_c_entry_points_every_rank_of_a_virtual_world(orc, 8, [1200] * 8)
    t.test_c_entry_points_every_rank_of_a_virtual_world(orc, 4, [2000, 1500, 2500, 1800])
    print("iteration", it, "ok", flush=True)
