"""The multi-rank HIP path as two real processes on one GPU: `bench.py --gpus 2 --rehearse-on-one-gpu --check-frame` under
torch.distributed.run -- both ranks render their Z-order tile share of the frame on device 0 (libprgpu: tile ownership, resident-pixel
scheduling of a half share), the frame is summed over gloo (staged through host memory; RCCL refuses two ranks on one device) and rank 0
compares it with a one-rank render of the same iterations.  What `tests/test_distributed_gloo.py` checks with the CPU checker as renderer,
here with the product; the RCCL transport itself runs in `test_a_frame_may_be_reduced_again_after_more_iterations` (one rank) and in the
driver's multi-GPU job."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_of_the_hip_path_on_one_device_sum_to_the_one_rank_frame():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--width", "640", "--height", "384", "--triangles", "50000",
           "--rehearse-on-one-gpu", "--check-frame", "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "tiles2" and out["config"]["tile_px"] == 64
    assert out["frame_check"] == {"xyz_equal": True, "samples_equal": True}, out["frame_check"]
    assert out["config"]["samples_per_step"] == 640 * 384          # the two shares cover the frame exactly once
    assert len(out["config"]["rank_render_ms_per_step"]["all"]) == 2
    # what a SCALE line needs to be read: who the collective spanned (null in the host-staged rehearsal: RCCL refuses two ranks on one
    # device), the reduce alone, and a roofline whose memory-side figure is this RANK's, not the full frame's (never above the peak)
    cfg = out["config"]
    assert {"rccl_ranks", "rccl_rank_of_root", "reduce_ms", "reduce_ms_in_timed_region", "collective"} <= set(cfg)
    assert cfg["collective"] == "torch.distributed.reduce" and cfg["rccl_ranks"] is None
    assert cfg["reduce_ms"] > 0.0
    rl = out["roofline"]
    assert rl["kernel_organisation"] in ("throughput", "latency")
    assert rl["traffic"] is None or rl["traffic"] <= rl["peak"], rl
    assert rl["achieved"] <= rl["peak"]
