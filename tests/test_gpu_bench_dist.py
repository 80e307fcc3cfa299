"""bench.py's N > 1 code path (two accumulation buffers, a stream of its own for the exchange, the C-ABI's RCCL reduce) with ONE
rank on one GPU: RT_BENCH_FORCE_DIST=1 initialises torch.distributed (nccl = RCCL) and the rt_comm of a world of one, so the code the
driver's 8-GPU run goes through is kept green here.  A child process, because bench.py owns its process group."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("strong", [False, True], ids=["weak", "strong"])
def test_bench_distributed_path_with_one_rank(strong):
    env = dict(os.environ, RT_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533" if strong else "29532",
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "C2", "--spp", "16", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-sub"]
    if strong:
        cmd.append("--strong")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=280, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["unit"] == "Mrays/s"
    assert line["value"] > 0 and line["value"] == line["value"] and line["ms_per_step"] > 0
    assert line["scaling"] == ("strong" if strong else "weak")
    assert "RCCL reduce" in line["config"]["sharding"] or line["config"]["sharding"] == "single GPU"
