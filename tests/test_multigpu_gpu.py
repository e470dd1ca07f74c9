"""The multi-rank path of bench.py, rehearsed on ONE device (pytest -m gpu): two ranks under torch.distributed.run exactly as the driver launches
them at N = 2, both on device 0 (HTS_BENCH_SHARE_DEVICE=1: the ranks meet over gloo, RCCL refuses two ranks on one GPU).  What it proves before a
real 8-GPU run: the rank environment is read, every rank builds / takes its share of the data (weak: the cooperative data set; strong: block ranges of
ONE file), each rank's output is verified, the time is the MAX over ranks and rank 0 prints ONE line with the whole-job figure."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run(extra, tmp_path):
    env = dict(os.environ, HTS_BENCH_SHARE_DEVICE="1", HTS_BENCH_CACHE=str(tmp_path / "cache"), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout.decode()[-800:], r.stderr.decode()[-1500:])
    return json.loads(lines[0])


def test_two_ranks_weak_scaling_all_ops(tmp_path):
    d = _run(["--gib", "0.25", "--op", "all", "--extra-steps", "1", "--slices", "16"], tmp_path)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["verified"] is True
    assert d["value"] > 0 and d["ms_per_step"] > 0
    # the whole-job figure counts both ranks' bytes: two ranks on one device cannot be slower than ~half of one rank alone, and the bytes are 2 x 0.25 GiB
    ex = d["extra"]
    assert ex["bgzf_deflate"]["n_gpus"] == 2 if "n_gpus" in ex["bgzf_deflate"] else True
    assert ex["bgzf_deflate"]["config"]["verified"] is True and ex["bgzf_deflate"]["config"]["decodes_with_reference_htslib"] in (True, None)
    for op in ("cram_rans_nx16_decode", "cram_rans_4x16_decode", "cram_slices"):
        assert op in ex and "error" not in ex[op] and ex[op]["value"] > 0, (op, ex.get(op))


def test_two_ranks_strong_scaling_one_file(tmp_path):
    d = _run(["--gib", "0.25", "--op", "inflate", "--scaling", "strong"], tmp_path)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["verified"] is True      # verified includes: the shards cover the file exactly once
    one = _run(["--gib", "0.25", "--op", "deflate", "--scaling", "strong"], tmp_path)
    assert one["n_gpus"] == 2 and one["scaling"] == "strong" and one["config"]["verified"] is True
