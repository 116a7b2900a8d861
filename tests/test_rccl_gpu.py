"""RCCL executed for real, on the one GPU a build box has: world size 1 on the `nccl` backend in a fresh process
(tests/tools/rccl_world1.py).  Proves that RCCL loads next to libpcsaft_hip.so, that the hand-off between the kernels'
stream and the collective's stream is ordered (the gathered values equal the kernels' outputs, for the plain and the
chunk-overlapped schedule of bench.py) and that the message-size logic of feos_torch_amd.dist runs on the real backend
(the reference's only parallelism is rayon over rows, src/pcsaft.rs:86-92).  No scaling number comes out of this:
multi-GPU throughput stays unmeasured until a multi-GPU node runs bench.py --gpus N."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_world1_allgather_matches_kernels(hip_lib):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "rccl_world1.py"), str(port)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["same_p"] and out["same_status"], out
    assert out["chunked_p"] and out["chunked_status"], out
    assert any("rccl" in name.lower() for name in out["libs"]), out  # the collective ran in RCCL, not in a stub
    assert any("libpcsaft_hip" in name for name in out["libs"]), out
