"""The N-rank path on the GPU box: bench.py starts its own ranks.

  * RCCL, world size 2: needs two GPUs (skipped on the 1-GPU box; the driver's multi-GPU tier runs it).
  * the SyncBN variant of BASELINE configs[3] (configs/sunrgbd/danet_r101_depthdeform.py:13,20,26:
    norm_cfg = SyncBN) likewise needs two GPUs: torch's SyncBatchNorm refuses CPU tensors and its own collectives cannot
    be drained for gloo from outside.
  * gloo, world size 2, both ranks on the ONE GPU: the rehearsal of the whole path (spawn, env contract, the gradient
    exchange after / from inside backward) for the layer and the plain-BN model workload.
"""
import json
import os
import signal
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(*argv, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    # its own process group: if the job hangs, the ranks it spawned go down with it (killing only the parent would
    # leave them on the GPU)
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        stdout, stderr = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)       # the group this call started, by its own id
        proc.communicate()
        raise
    assert proc.returncode == 0, stderr[-2000:]
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout            # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_bench_gpus2_runs_two_ranks_on_rccl():
    line = _bench("--gpus", "2", "--steps", "3", "--warmup", "1")
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2"
    assert line["allreduce"]["distinct_devices"] == 2 and len(line["devices"]) == 2
    assert line["allreduce"]["configs3_gradient"]["bytes"] == 276_000_000
    assert line["allreduce"]["configs3_gradient"]["bus_GBps"] > 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_bench_gpus2_syncbn_model_workload_on_rccl():
    """The SyncBN variant of BASELINE configs[3] at a reduced input size: SyncBatchNorm's all_gather / all_reduce and
    the bucketed, backward-overlapped gradient all-reduce, all over RCCL."""
    line = _bench("--gpus", "2", "--workload", "r50", "--syncbn", "--batch", "1", "--hw", "97", "129", "--steps", "2",
                  "--warmup", "1")
    assert line["n_gpus"] == 2 and line["config"]["norm"] == "SyncBN"
    assert line["config"]["parallelism"] == "dp2" and line["allreduce"]["distinct_devices"] == 2
    assert line["allreduce"]["gradient"]["bytes"] > 100e6          # DDCN-R50 + DANet head: ~50 M fp32 parameters
    assert line["final_loss"] == line["final_loss"] and line["final_loss"] > 0      # finite


def test_bench_gpus2_rehearsal_on_one_gpu_over_gloo():
    """The N-rank path end to end on the one GPU this box has: bench.py starts its two ranks itself, both run the layer
    on the same device, the gradient goes through gloo (dist_utils drains the stream first: two ranks sharing a GPU
    otherwise wait for each other inside gloo), rank 0 prints the one line."""
    line = _bench("--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", timeout=100)
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and len(line["devices"]) == 2
    assert line["allreduce"]["backend"].startswith("gloo") and line["allreduce"]["distinct_devices"] == 1
    assert line["allreduce"]["grad_weight"]["bytes"] == 512 * 512 * 9 * 4 and line["value"] > 0


def test_bench_gpus2_model_workload_rehearsal_over_gloo():
    """configs[3]'s training step with two ranks on the one GPU: bucketed gradient all-reduce issued from inside backward
    (OverlappedGradReducer, async collectives) over gloo, reduced input size.  (SyncBatchNorm's own collectives cannot be
    drained from here, so the SyncBN variant needs RCCL and two GPUs: the test above.)"""
    line = _bench("--gpus", "2", "--backend", "gloo", "--workload", "r50", "--batch", "1", "--hw", "97", "129",
                  "--steps", "2", "--warmup", "1", timeout=110)
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["config"]["norm"] == "BN"
    assert line["config"]["allreduce"].startswith("32 MiB buckets")
    assert line["allreduce"]["backend"].startswith("gloo") and line["allreduce"]["gradient"]["bytes"] > 100e6
    # every bucket size of the reducer is timed alone: the sizes add up to the whole gradient
    buckets = line["allreduce"]["buckets"]
    assert sum(b["bytes"] * b["count"] for b in buckets) == line["allreduce"]["gradient"]["bytes"]
    assert all(b["ms"] > 0 and b["bus_GBps"] > 0 for b in buckets)
    assert line["final_loss"] == line["final_loss"] and line["final_loss"] > 0


def test_bench_single_gpu_line_has_the_contract_fields():
    line = _bench("--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    assert line["n_gpus"] == 1 and line["metric"] == "DDCN op fwd+bwd GB/s" and line["dtype"] == "f32"
    assert line["roofline"]["bound"] == "mfma" and 0 < line["roofline"]["frac"] <= 1
    assert "traffic_source" in line["roofline"] and line["gather_GBps_executed"] > 0


def test_train_cli_runs_and_resumes(tmp_path):
    """tools/train.py (the reference's tools/train.py:15-109 shape): two iterations, a checkpoint, a resumed run."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    base = [sys.executable, os.path.join(ROOT, "tools", "train.py"), "danet_r50_depthdeform", "--crop", "65", "97",
            "--imgs_per_gpu", "1", "--log_interval", "1", "--seed", "3"]
    r = subprocess.run(base + ["--iters", "2", "--max_iters", "3", "--work_dir", str(tmp_path)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    logs = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert [l["iter"] for l in logs if "iter" in l] == [1, 2] and logs[-1]["checkpoint"].endswith("iter_2.pth")
    r2 = subprocess.run(base + ["--iters", "1", "--resume_from", logs[-1]["checkpoint"]], env=env, capture_output=True,
                        text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    logs2 = [json.loads(l) for l in r2.stdout.splitlines() if l.startswith("{")]
    assert logs2[0]["iter"] == 3 and logs2[0]["loss"] == logs2[0]["loss"]
    # the resumed run continues on the first run's polynomial (schedule length kept in the checkpoint, as mmcv's runner
    # keeps total_epochs): its learning rate at iteration 3 is the one of an uninterrupted 3-iteration run
    r3 = subprocess.run(base + ["--iters", "3"], env=env, capture_output=True, text=True, timeout=600)
    assert r3.returncode == 0, r3.stderr[-2000:]
    logs3 = [json.loads(l) for l in r3.stdout.splitlines() if l.startswith("{") and "iter" in l]
    assert [l["iter"] for l in logs3] == [1, 2, 3]
    assert logs2[0]["lr"] == pytest.approx(logs3[2]["lr"], rel=1e-12)
    assert [l["lr"] for l in logs if "iter" in l] == pytest.approx([l["lr"] for l in logs3[:2]], rel=1e-12)
    assert logs3[2]["lr"] == pytest.approx(0.00025 * (1 - 2 / 3) ** 0.9, rel=1e-9)
