"""The HIP IPC calls behind CLC_MC_PEER_COPY (coloc_amd/csrc/multicam.hip open_peers: hipIpcGetMemHandle on the arena, the handle sent to the
peers, hipIpcOpenMemHandle(hipIpcMemLazyEnablePeerAccess) there, device-to-device copies through the mapped pointer, hipIpcCloseMemHandle)
between TWO PROCESSES -- on the one GPU the build loop has, so same-device mapping instead of a peer's, but the same calls, flags and
HSA_ENABLE_IPC_MODE_LEGACY=0 environment the multi-GPU run depends on.  The communicator-side of that path (handle exchange and fence
through ncclAllGather) is covered by test_one_rank_communicator_drives_the_rccl_abi."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CHILD = r'''
import ctypes as C, sys
import numpy as np
hip = C.CDLL("libamdhip64.so")
def chk(rc, what):
    if rc != 0:
        raise SystemExit("%s failed: %d" % (what, rc))
handle = bytes.fromhex(sys.argv[1]); n = int(sys.argv[2])
chk(hip.hipSetDevice(0), "hipSetDevice")
class H(C.Structure):
    _fields_ = [("reserved", C.c_char * 64)]
h = H(); C.memmove(C.byref(h), handle, 64)
p = C.c_void_p()
hip.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), H, C.c_uint]
chk(hip.hipIpcOpenMemHandle(C.byref(p), h, 1), "hipIpcOpenMemHandle")        # hipIpcMemLazyEnablePeerAccess
mine = C.c_void_p(); chk(hip.hipMalloc(C.byref(mine), C.c_size_t(n)), "hipMalloc")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
chk(hip.hipMemcpy(mine, p, n, 3), "D2D from the mapped arena")                # read the parent's block through the mapping
host = np.zeros(n, np.uint8); chk(hip.hipMemcpy(host.ctypes.data, mine, n, 2), "D2H")
want = (np.arange(n, dtype=np.uint32) * 7 + 3).astype(np.uint8)
assert np.array_equal(host, want), "child read other bytes than the parent wrote"
back = (255 - want).astype(np.uint8)
chk(hip.hipMemcpy(mine, back.ctypes.data, n, 1), "H2D"); chk(hip.hipMemcpy(p, mine, n, 3), "D2D into the mapped arena")   # the peer copy
chk(hip.hipDeviceSynchronize(), "sync"); chk(hip.hipIpcCloseMemHandle(p), "hipIpcCloseMemHandle")
print("child ok")
'''


def test_ipc_handle_round_trip_between_two_processes():
    import ctypes as C
    import torch
    if os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "") != "0":
        pytest.skip("HSA_ENABLE_IPC_MODE_LEGACY=0 is not set: this pool's driver only supports dmabuf IPC")
    torch.cuda.init()                      # this process owns the GPU through PyTorch's HIP runtime (as bench.py does)
    from coloc_amd import abi
    abi.load_library()                     # maps the same libamdhip64 torch uses
    hip = C.CDLL("libamdhip64.so")
    n = 640 * 1024                         # one camera's descriptor block at 10k keypoints
    buf = torch.from_numpy((np.arange(n, dtype=np.uint32) * 7 + 3).astype(np.uint8)).cuda()
    torch.cuda.synchronize()

    class H(C.Structure):
        _fields_ = [("reserved", C.c_char * 64)]
    h = H()
    hip.hipIpcGetMemHandle.argtypes = [C.POINTER(H), C.c_void_p]
    # the handle of a pointer inside a torch allocation refers to the allocation's base: use a buffer of our own
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(n)) == 0
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(p, C.c_void_p(buf.data_ptr()), n, 3) == 0
    assert hip.hipIpcGetMemHandle(C.byref(h), p) == 0
    out = subprocess.run([sys.executable, "-c", CHILD, bytes(h)[:64].hex(), str(n)], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ))
    assert out.returncode == 0 and "child ok" in out.stdout, (out.stdout + out.stderr)[-2000:]
    assert hip.hipDeviceSynchronize() == 0
    got = torch.empty(n, dtype=torch.uint8, device="cuda")
    assert hip.hipMemcpy(C.c_void_p(got.data_ptr()), p, n, 3) == 0
    torch.cuda.synchronize()
    want = 255 - (np.arange(n, dtype=np.uint32) * 7 + 3).astype(np.uint8)
    assert np.array_equal(got.cpu().numpy(), want)      # the child's peer copy landed in this process's buffer
    assert hip.hipFree(p) == 0
